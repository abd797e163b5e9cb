// Mutual nearest-neighbour descriptor matcher on gfx950: D0 . D1^T on
// v_mfma_f32_32x32x16_f16 with the row top-2 reduction fused into the GEMM epilogue (the
// N x M similarity matrix never exists in memory), then ratio / distance / mutual tests.
//
// Replaces (file:line in the reference):
//   hloc/matchers/nearest_neighbor.py:6-16 find_nn, :19-24 mutual_check, :38-57 _forward
//   it_loc/matcher.py:91-119 Matcher.forward, :122-130 mutual_nn_matcher, :165-194 mutual_nn_ratio_matcher
//
// Orientation: the side being REDUCED over ("a": candidates j) is the MFMA A operand (rows),
// the side being KEPT ("b": queries i) is the B operand (columns).  In the 32x32 C/D layout
// (col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)) a lane then holds 16 different candidates
// of ONE query, so the running top-2 is lane-local; one cross-half merge at the end.
// The reverse direction (column top-2 of the similarity matrix) is the same kernel with the
// operands swapped.
#include "sfd2_internal.h"
#include <math.h>
#include <stdlib.h>

#define NT 256
#define KD 128        // descriptor dimension (nets/sfd2.py:260 outdim=128)
#define TA 64         // candidate rows per LDS stage
#define APITCH 136    // fp16 per staged row: 128 + 8 pad (272 B)

// ---------------------------------------------------------------- operand preparation
// hi = fp16(v);  lo = fp16((v - hi) * 2^11)  so that  v ~= hi + lo * 2^-11  to ~2^-22 relative.
__global__ __launch_bounds__(NT)
void match_prep_kernel(const void *__restrict__ src, int n, int n_src, const int *__restrict__ rows, int dim, int dtype,
                       int layout, half_t *__restrict__ hi, half_t *__restrict__ lo)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * KD) return;
    const int i = (int)(t / KD), d = (int)(t % KD);
    double v = 0.0;
    if (d < dim) {
        const int r = rows ? rows[i] : i;   // gathered subset (feature_matching's 3D-point mask, localize_cv2.py:531-534)
        const size_t off = layout == 0 ? (size_t)r * dim + d : (size_t)d * n_src + r;
        if (dtype == 0) v = (double)reinterpret_cast<const float *>(src)[off];
        else if (dtype == 1) v = reinterpret_cast<const double *>(src)[off];
        else v = (double)(float)reinterpret_cast<const half_t *>(src)[off];
    }
    const half_t h = (half_t)(float)v;
    hi[t] = h;
    if (lo) lo[t] = (half_t)(float)((v - (double)(float)h) * 2048.0);
}

void launch_match_prep(hipStream_t st, const void *src, int n, int n_src, const int *rows, int dim, int dtype, int layout,
                       half_t *hi, half_t *lo)
{
    const size_t tot = (size_t)n * KD;
    if (tot == 0) return;
    hipLaunchKernelGGL(match_prep_kernel, dim3((unsigned)((tot + NT - 1) / NT)), dim3(NT), 0, st, src, n, n_src, rows, dim,
                       dtype, layout, hi, lo);
}

// ---------------------------------------------------------------- fused GEMM + top-2
__device__ __forceinline__ void top2_update(float v, int j, float &b1, float &b2, int &i1)
{
    // strict '>' keeps the first (lowest) index among equal maxima, like torch.max / our tie rule
    b2 = __builtin_amdgcn_fmed3f(b1, b2, v);
    i1 = v > b1 ? j : i1;
    b1 = fmaxf(b1, v);
}

template <bool USE_LO>
__global__ __launch_bounds__(NT, 2)
void match_top2_kernel(const MatchJob *__restrict__ jobs, int splits)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t *As = reinterpret_cast<half_t *>(smem);                 // [2][TA][APITCH] hi
    half_t *Al = As + 2 * TA * APITCH;                             // [2][TA][APITCH] lo (USE_LO)
    const MatchJob job = jobs[blockIdx.z];
    const int na = job.na, nb = job.nb;
    const int i_base = blockIdx.x * 128;
    if (i_base >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lcol = lane & 31, lk = (lane >> 5) * 8;

    // candidate range of this split, multiple of 32 so sub-tiles never straddle splits
    int chunk = (na + splits - 1) / splits;
    chunk = (chunk + 31) & ~31;
    const int ja0 = blockIdx.y * chunk;
    int ja1 = ja0 + chunk;
    if (ja1 > na) ja1 = na;

    const int my_i = i_base + wave * 32 + lcol;
    h8_t bh[8], bl[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        h8_t z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (half_t)0.0f;
        bh[ks] = z;
        bl[ks] = z;
        if (my_i < nb) {
            bh[ks] = *reinterpret_cast<const h8_t *>(job.b_hi + (size_t)my_i * KD + ks * 16 + lk);
            if (USE_LO) bl[ks] = *reinterpret_cast<const h8_t *>(job.b_lo + (size_t)my_i * KD + ks * 16 + lk);
        }
    }

    float b1 = -INFINITY, b2 = -INFINITY;
    int i1 = 0;

    if (ja0 < ja1) {
        const int nst = (ja1 - ja0 + TA - 1) / TA;
        uint4 rh[4], rl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { rh[i] = make_uint4(0, 0, 0, 0); rl[i] = make_uint4(0, 0, 0, 0); }
        // piece p (0..1023): row = p >> 4, part = p & 15 (16 bytes each)
#define LOAD_A(stage_)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
        const int p = tid + i * NT, row = p >> 4, part = p & 15;                                       \
        const int ja = ja0 + (stage_)*TA + row;                                                        \
        rh[i] = make_uint4(0, 0, 0, 0);                                                                \
        rl[i] = make_uint4(0, 0, 0, 0);                                                                \
        if (ja < ja1) {                                                                                \
            rh[i] = *reinterpret_cast<const uint4 *>(job.a_hi + (size_t)ja * KD + part * 8);           \
            if (USE_LO) rl[i] = *reinterpret_cast<const uint4 *>(job.a_lo + (size_t)ja * KD + part * 8); \
        }                                                                                              \
    }
#define STORE_A(buf_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
        const int p = tid + i * NT, row = p >> 4, part = p & 15;                                       \
        *reinterpret_cast<uint4 *>(As + ((buf_)*TA + row) * APITCH + part * 8) = rh[i];                \
        if (USE_LO) *reinterpret_cast<uint4 *>(Al + ((buf_)*TA + row) * APITCH + part * 8) = rl[i];    \
    }
        LOAD_A(0)
        STORE_A(0)
        __syncthreads();
        for (int s = 0; s < nst; ++s) {
            const int buf = s & 1;
            const bool has_next = s + 1 < nst;
            if (has_next) { LOAD_A(s + 1) }
#pragma unroll
            for (int sub = 0; sub < TA / 32; ++sub) {
                f32x16_t acc, accx;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; accx[r] = 0.0f; }
                const half_t *arow = As + (buf * TA + sub * 32 + lcol) * APITCH + lk;
                const half_t *arow_l = Al + (buf * TA + sub * 32 + lcol) * APITCH + lk;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const h8_t ah = *reinterpret_cast<const h8_t *>(arow + ks * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], acc, 0, 0, 0);
                    if (USE_LO) {
                        const h8_t al = *reinterpret_cast<const h8_t *>(arow_l + ks * 16);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], accx, 0, 0, 0);
                        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], accx, 0, 0, 0);
                    }
                }
                const int jbase = ja0 + s * TA + sub * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int j = jbase + (r & 3) + 8 * (r >> 2);
                    float v = acc[r];
                    if (USE_LO) v = v + accx[r] * (1.0f / 2048.0f);
                    if (j >= ja1) v = -INFINITY;
                    top2_update(v, j, b1, b2, i1);
                }
            }
            if (has_next) { STORE_A(buf ^ 1) }
            __syncthreads();
        }
#undef LOAD_A
#undef STORE_A
    }
    // merge the two half-waves (same query, interleaved candidate rows)
    const float c1 = __shfl_xor(b1, 32), c2 = __shfl_xor(b2, 32);
    const int j1 = __shfl_xor(i1, 32);
    float n1v, n2v;
    int n1i;
    if (c1 > b1 || (c1 == b1 && j1 < i1)) { n1v = c1; n1i = j1; n2v = fmaxf(b1, c2); }
    else { n1v = b1; n1i = i1; n2v = fmaxf(b2, c1); }
    if (lane < 32 && my_i < nb) {
        const size_t o = (size_t)blockIdx.y * nb + my_i;
        job.part_v1[o] = n1v;
        job.part_v2[o] = n2v;
        job.part_i1[o] = n1i;
    }
}

// ---------------------------------------------------------------- second generation (fp16 operands)
// 64 queries per wave (each candidate fragment read from LDS feeds two MFMAs), candidates staged by
// direct-to-LDS copies (global_load_lds_dwordx4) into a lane-linear image of 256-byte rows whose
// 16-byte slots are XOR-swizzled with (row & 15); top-2 or top-1 tracking chosen at compile time.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define TA2 64        // candidate rows per LDS stage in the v2 kernel (16 KB; 128 measured slower: fewer blocks per CU)
template <bool NEED2, int ABL = 0>   // ABL: timing ablations only (wrong results): 1 = no arg-max epilogue
__global__ __launch_bounds__(NT, 2)   // 140 VGPRs -> 3 blocks per CU; forcing 4 (128 VGPRs) spills and measured slower
void match_top2_v2_kernel(const MatchJob *__restrict__ jobs, int splits, const half_t *__restrict__ zero_page)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][TA2][256 B]
    const MatchJob job = jobs[blockIdx.z];
    const int na = job.na, nb = job.nb;
    const int i_base = blockIdx.x * 256;
    if (i_base >= nb) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 31, lhi = lane >> 5;

    int chunk = (na + splits - 1) / splits;
    chunk = (chunk + 31) & ~31;
    const int ja0 = blockIdx.y * chunk;
    int ja1 = ja0 + chunk;
    if (ja1 > na) ja1 = na;

    h8_t bq[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = i_base + wave * 64 + t * 32 + lcol;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            h8_t z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (half_t)0.0f;
            bq[t][ks] = z;
            if (qi < nb) bq[t][ks] = *reinterpret_cast<const h8_t *>(job.b_hi + (size_t)qi * KD + ks * 16 + lhi * 8);
        }
    }
    float b1[2] = {-INFINITY, -INFINITY}, b2[2] = {-INFINITY, -INFINITY};
    int i1[2] = {0, 0};
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    // top-1 mode: running maximum with the register index (15 - r) packed into the 4 low mantissa
    // bits (relative perturbation <= 2^-19, far below the fp16-operand error), plus the sub-tile id
    float bp[2] = {-INFINITY, -INFINITY};
    int sid[2] = {0, 0};

    if (ja0 < ja1) {
        const int nst = (ja1 - ja0 + TA2 - 1) / TA2;
        // staging: TA2/4 one-KB chunks (4 rows each) per stage, TA2/16 per wave
        const int srow = lane >> 4;                                  // row within the chunk
#define ISSUE_A(stage_, buf_)                                                                            \
    _Pragma("unroll") for (int c = 0; c < TA2 / 16; ++c) {                                               \
        const int row = (wave * (TA2 / 16) + c) * 4 + srow;                                                       \
        const int slot = (lane & 15) ^ (row & 15);                                                       \
        const int ja = ja0 + (stage_)*TA2 + row;                                                          \
        const half_t *src = ja < ja1 ? job.a_hi + (size_t)ja * KD + slot * 8 : zero_page + (lane & 3) * 8; \
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src,                                              \
                                         (lds_void_t *)(smem + (buf_)*TA2 * 256 + (wave * (TA2 / 16) + c) * 1024), 16, 0, 0); \
    }
        ISSUE_A(0, 0)
        SFD2_BARRIER_DRAIN();
        for (int s = 0; s < nst; ++s) {
            const int buf = s & 1;
            if (s + 1 < nst) { ISSUE_A(s + 1, buf ^ 1) }
#pragma unroll
            for (int sub = 0; sub < TA2 / 32; ++sub) {
                f32x16_t acc0 = zero16, acc1 = zero16;
                const int row = sub * 32 + lcol;
                const unsigned char *arow = smem + (buf * TA2 + row) * 256;
                const int sw = row & 15;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const h8_t a = *reinterpret_cast<const h8_t *>(arow + (((ks * 2 + lhi) ^ sw) << 4));
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[0][ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[1][ks], acc1, 0, 0, 0);
                }
                const int jbase = ja0 + s * TA2 + sub * 32 + 4 * lhi;
                const bool full = ja0 + s * TA2 + sub * 32 + 32 <= ja1;   // wave-uniform: no row of this sub-tile is padding
                if (ABL == 1) {
                    bp[0] = fmaxf(bp[0], acc0[0] + acc0[5] + acc0[10] + acc0[15]);
                    bp[1] = fmaxf(bp[1], acc1[0] + acc1[5] + acc1[10] + acc1[15]);
                } else if (!NEED2 && full) {
                    // one v_and_or per element packs (15 - r) under the value, a max tree then yields the
                    // maximum AND its register; per-element index scans would make this kernel VALU-bound
                    // (measured 11.8 VALU instructions per MFMA with them)
                    const int sub_id = s * (TA2 / 32) + sub;
                    float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p0 = __uint_as_float((__float_as_uint(acc0[r]) & 0xFFFFFFF0u) | (unsigned)(15 - r));
                        const float p1 = __uint_as_float((__float_as_uint(acc1[r]) & 0xFFFFFFF0u) | (unsigned)(15 - r));
                        m0 = fmaxf(m0, p0);
                        m1 = fmaxf(m1, p1);
                    }
                    sid[0] = m0 > bp[0] ? sub_id : sid[0]; bp[0] = fmaxf(bp[0], m0);
                    sid[1] = m1 > bp[1] ? sub_id : sid[1]; bp[1] = fmaxf(bp[1], m1);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = jbase + (r & 3) + 8 * (r >> 2);
                        const bool in = j < ja1;
                        const float v0 = in ? acc0[r] : -INFINITY, v1 = in ? acc1[r] : -INFINITY;
                        if (NEED2) {
                            top2_update(v0, j, b1[0], b2[0], i1[0]);
                            top2_update(v1, j, b1[1], b2[1], i1[1]);
                        } else {
                            i1[0] = v0 > b1[0] ? j : i1[0]; b1[0] = fmaxf(b1[0], v0);
                            i1[1] = v1 > b1[1] ? j : i1[1]; b1[1] = fmaxf(b1[1], v1);
                        }
                    }
                }
            }
            SFD2_BARRIER_DRAIN();
        }
#undef ISSUE_A
    }
    if (!NEED2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned int bits = __float_as_uint(bp[t]);
            const float v = __uint_as_float(bits & 0xFFFFFFF0u);
            const int r = 15 - (int)(bits & 15u);
            const int j = ja0 + sid[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            // the packed path and the masked tail path (b1/i1) both ran: keep the better, lower index on ties
            if (bp[t] != -INFINITY && (v > b1[t] || (v == b1[t] && j < i1[t]))) { b1[t] = v; i1[t] = j; }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float c1 = __shfl_xor(b1[t], 32), c2 = __shfl_xor(b2[t], 32);
        const int j1 = __shfl_xor(i1[t], 32);
        float n1v, n2v;
        int n1i;
        if (c1 > b1[t] || (c1 == b1[t] && j1 < i1[t])) { n1v = c1; n1i = j1; n2v = fmaxf(b1[t], c2); }
        else { n1v = b1[t]; n1i = i1[t]; n2v = fmaxf(b2[t], c1); }
        const int qi = i_base + wave * 64 + t * 32 + lcol;
        if (lane < 32 && qi < nb) {
            const size_t o = (size_t)blockIdx.y * nb + qi;
            job.part_v1[o] = n1v;
            job.part_v2[o] = NEED2 ? n2v : -INFINITY;
            job.part_i1[o] = n1i;
        }
    }
}

// ---------------------------------------------------------------- single-GEMM mutual NN: match_mutual_kernel.hip
#define MQ_NEG (-0x1p100f)
void launch_match_mutual_gemm(hipStream_t st, const MatchJob2 *jobs_dev, int npairs, int max_n0, int splits, int fwd_ids);
int match_mutual_strip(void);

// merges the partials: forward [splits][n0] (value, index -- fwd_ids = 0: value only, the index is left at -1 for
// match_mutual_claim_kernel), reverse [ceil(n0 / strip)][n1] packed (value | query id bits)
__global__ __launch_bounds__(NT)
void match_mutual_reduce_kernel(const MatchJob2 *__restrict__ jobs, const MatchFinal *__restrict__ fins, int splits, int strip, int fwd_ids)
{
    const MatchJob2 job = jobs[blockIdx.y];
    const MatchFinal f = fins[blockIdx.y];
    const int dir = blockIdx.z;
    const int n = dir == 0 ? job.n0 : job.n1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float b1 = -INFINITY;
    int bi = 0;
    if (dir == 0) {
        for (int s = 0; s < splits; ++s) {
            const float c1 = job.part_v1[(size_t)s * n + i];
            if (c1 > b1) { b1 = c1; if (fwd_ids) bi = job.part_i1[(size_t)s * n + i]; }
        }
        if (!fwd_ids) bi = -1;
        f.red_f[i] = b1;
        f.red_f[(size_t)n + i] = -INFINITY;
        reinterpret_cast<int *>(f.red_f)[2 * (size_t)n + i] = bi;
    } else {
        const int nstrip = (job.n0 + strip - 1) / strip;
        // strips compare by VALUE (id bits masked off): among equal values the lower strip = the lower query index wins
        float best = MQ_NEG;
        unsigned int bkey = __float_as_uint(MQ_NEG);
        int bs = 0;
        int sidx = 0;
        for (; sidx + 8 <= nstrip; sidx += 8) {            // 8 independent loads in flight per thread (13 MB per 50 pairs)
            float k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) k[u] = job.rkeys[(size_t)(sidx + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = __uint_as_float(__float_as_uint(k[u]) & 0xFFFFFF00u);
                if (v > best) { best = v; bkey = __float_as_uint(k[u]); bs = sidx + u; }   // ascending strips, strict '>'
            }
        }
        for (; sidx < nstrip; ++sidx) {
            const float k = job.rkeys[(size_t)sidx * n + i];
            const float v = __uint_as_float(__float_as_uint(k) & 0xFFFFFF00u);
            if (v > best) { best = v; bkey = __float_as_uint(k); bs = sidx; }
        }
        const unsigned int bits = bkey;
        // id bits = 255 - query index within the block's 256-query strip (match_mutual_kernel.hip)
        const bool any = best > MQ_NEG;
        f.red_r[i] = any ? __uint_as_float(bits & 0xFFFFFF00u) : -INFINITY;
        f.red_r[(size_t)n + i] = -INFINITY;
        reinterpret_cast<int *>(f.red_r)[2 * (size_t)n + i] = any ? bs * strip + 255 - (int)(bits & 255u) : 0;
    }
}

// Mutual modes without forward ids: (i, j) is a match iff sim[i][j] is the maximum of column j -- i = the reverse direction's
// query id for j -- AND of row i: the row maximum F[i] (raw accumulator value) equals the column maximum R[j] (the same
// accumulator value with its 8 low mantissa bits replaced by id bits).  The forward index of query i becomes the LOWEST
// such j (unsigned atomicMin over an array initialised to 0xFFFFFFFF = -1: torch's argmax takes the first maximum,
// nearest_neighbor.py:8 / it_loc/matcher.py:124); queries nobody claims keep -1 and match_decide_kernel leaves them unmatched.
__global__ __launch_bounds__(NT)
void match_mutual_claim_kernel(const MatchFinal *__restrict__ fins)
{
    const MatchFinal f = fins[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= f.n1) return;
    const float r = f.red_r[j];
    if (!(r > -INFINITY)) return;
    const int i = reinterpret_cast<const int *>(f.red_r)[2 * (size_t)f.n1 + j];
    if ((__float_as_uint(f.red_f[i]) & 0xFFFFFF00u) == __float_as_uint(r))
        atomicMin(reinterpret_cast<unsigned int *>(f.red_f) + 2 * (size_t)f.n0 + i, (unsigned int)j);
}

void launch_match_mutual(hipStream_t st, const MatchJob2 *jobs_dev, const MatchFinal *fins_dev, int npairs, int max_n0,
                         int max_n1, int splits, int fwd_ids)
{
    if (npairs <= 0 || max_n0 <= 0) return;
    launch_match_mutual_gemm(st, jobs_dev, npairs, max_n0, splits, fwd_ids);
    const int max_n = max_n0 > max_n1 ? max_n0 : max_n1;
    hipLaunchKernelGGL(match_mutual_reduce_kernel, dim3((max_n + NT - 1) / NT, npairs, 2), dim3(NT), 0, st, jobs_dev, fins_dev, splits,
                       match_mutual_strip(), fwd_ids);
    if (!fwd_ids && max_n1 > 0)
        hipLaunchKernelGGL(match_mutual_claim_kernel, dim3((max_n1 + NT - 1) / NT, npairs), dim3(NT), 0, st, fins_dev);
}

void launch_match_top2(hipStream_t st, const MatchJob *jobs_dev, int njobs, int max_nb, int splits, int use_lo,
                       int need_top2, const half_t *zero_page)
{
    if (njobs <= 0 || max_nb <= 0) return;
    static bool attr_done = false;
    const size_t lds_hi = (size_t)2 * TA * APITCH * sizeof(half_t);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_top2_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_hi));
        attr_done = true;
    }
    if (!use_lo && zero_page && sfd2_env("SFD2_MATCH_V1") == nullptr) {
        const dim3 grid((max_nb + 255) / 256, splits, njobs);
        const size_t lds = (size_t)2 * TA2 * 256;
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_top2_v2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_top2_v2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#ifdef SFD2_EXPERIMENTS
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_top2_v2_kernel<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
#endif
            attr2 = true;
        }
        if (need_top2) hipLaunchKernelGGL(match_top2_v2_kernel<true>, grid, dim3(NT), lds, st, jobs_dev, splits, zero_page);
#ifdef SFD2_EXPERIMENTS
        else if (sfd2_env("SFD2_MATCH_ABLATE")) hipLaunchKernelGGL((match_top2_v2_kernel<false, 1>), grid, dim3(NT), lds, st, jobs_dev, splits, zero_page);
#endif
        else hipLaunchKernelGGL(match_top2_v2_kernel<false>, grid, dim3(NT), lds, st, jobs_dev, splits, zero_page);
        return;
    }
    const dim3 grid((max_nb + 127) / 128, splits, njobs);
    if (use_lo) hipLaunchKernelGGL(match_top2_kernel<true>, grid, dim3(NT), 2 * lds_hi, st, jobs_dev, splits);
    else hipLaunchKernelGGL(match_top2_kernel<false>, grid, dim3(NT), lds_hi, st, jobs_dev, splits);
}

// ---------------------------------------------------------------- split merge + decisions
// red[0][i] = best similarity, red[1][i] = second best, red[2][i] = arg best (int bits)
__global__ __launch_bounds__(NT)
void match_reduce_kernel(const MatchFinal *__restrict__ fins, int splits)
{
    const MatchFinal f = fins[blockIdx.y];
    const int dir = blockIdx.z;
    const int n = dir == 0 ? f.n0 : f.n1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *v1 = dir == 0 ? f.f_v1 : f.r_v1;
    const float *v2 = dir == 0 ? f.f_v2 : f.r_v2;
    const int *i1 = dir == 0 ? f.f_i1 : f.r_i1;
    float *red = dir == 0 ? f.red_f : f.red_r;
    float b1 = -INFINITY, b2 = -INFINITY;
    int bi = 0;
    for (int s = 0; s < splits; ++s) {  // splits ascend in candidate index: strict '>' keeps the lowest
        const float c1 = v1[(size_t)s * n + i], c2 = v2[(size_t)s * n + i];
        const int ci = i1[(size_t)s * n + i];
        if (c1 > b1) { b2 = fmaxf(b1, c2); b1 = c1; bi = ci; }
        else { b2 = fmaxf(b2, c1); }
    }
    red[i] = b1;
    red[(size_t)n + i] = b2;
    reinterpret_cast<int *>(red)[2 * (size_t)n + i] = bi;
}

__device__ __forceinline__ bool hloc_pass(float s1, float s2, float ratio, float dist)
{
    const float d0 = 2.0f * (1.0f - s1), d1 = 2.0f * (1.0f - s2);     // find_nn: dist_nn = 2 * (1 - sim_nn)
    bool ok = true;
    if (ratio > 0.0f) ok = ok && (d0 <= (ratio * ratio) * d1);
    if (dist > 0.0f) ok = ok && (d0 <= dist * dist);
    return ok;
}
__device__ __forceinline__ float lowe_ratio(float s1, float s2)
{
    // it_loc/matcher.py:171-174: sqrt(2 - 2 sim), ratio = d0 / (d1 + 1e-8).  2 - 2 sim is clamped at 0
    // (the reference yields NaN -> "no match" when rounding pushes a self-similarity above 1).
    const float d0 = sqrtf(fmaxf(2.0f - 2.0f * s1, 0.0f)), d1 = sqrtf(fmaxf(2.0f - 2.0f * s2, 0.0f));
    return d0 / (d1 + 1e-8f);
}

__global__ __launch_bounds__(NT)
void match_decide_kernel(const MatchFinal *__restrict__ fins, int flavour, int mutual, float ratio, float dist)
{
    const MatchFinal f = fins[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.n0) return;
    const float s1 = f.red_f[i], s2 = f.red_f[(size_t)f.n0 + i];
    const int j = reinterpret_cast<const int *>(f.red_f)[2 * (size_t)f.n0 + i];
    long long m = -1;
    float score = 0.0f;
    if (f.n1 > 0 && j < 0) {
        // single-GEMM mutual modes (match_mutual_claim_kernel): no candidate has this query as its best one
        score = flavour == 0 ? (hloc_pass(s1, s2, ratio, dist) ? (s1 + 1.0f) / 2.0f : 0.0f) : s1;
    } else if (f.n1 > 0) {
        const float t1 = f.red_r[j], t2 = f.red_r[(size_t)f.n1 + j];
        const int back = reinterpret_cast<const int *>(f.red_r)[2 * (size_t)f.n1 + j];
        if (flavour == 0) {  // hloc NearestNeighbor
            const bool ok = hloc_pass(s1, s2, ratio, dist);
            score = ok ? (s1 + 1.0f) / 2.0f : 0.0f;   // scores0 come from the forward direction only
            m = ok ? j : -1;
            if (ok && mutual) {
                const bool ok_back = hloc_pass(t1, t2, ratio, dist);
                if (!(ok_back && back == i)) m = -1;
            }
        } else if (flavour == 1) {  // it_loc nnm
            score = s1;
            m = (back == i) ? j : -1;
        } else {  // it_loc nnr
            score = s1;
            const bool ok = (back == i) && lowe_ratio(s1, s2) <= ratio && lowe_ratio(t1, t2) <= ratio;
            m = ok ? j : -1;
        }
    }
    const long long mm = (m >= 0 && f.remap) ? (long long)f.remap[m] : m;   // back to unmasked indexing (localize_cv2.py:557-559)
    if (f.out16) {      // the casts of hloc/match_features.py:114,118 on the device: .short() (wraps) and .half() (round to nearest even)
        reinterpret_cast<short *>(f.matches0)[i] = (short)mm;
        reinterpret_cast<half_t *>(f.scores0)[i] = (half_t)score;
    } else {
        f.matches0[i] = mm;
        f.scores0[i] = score;
    }
}

void launch_match_decide(hipStream_t st, const MatchFinal *fin_dev, int npairs, int max_n, int flavour, int mutual,
                         float ratio, float dist)
{
    if (npairs <= 0 || max_n <= 0) return;
    hipLaunchKernelGGL(match_decide_kernel, dim3((max_n + NT - 1) / NT, npairs), dim3(NT), 0, st, fin_dev, flavour,
                       mutual, ratio, dist);
}

void launch_match_finalize(hipStream_t st, const MatchFinal *fin_dev, int npairs, int max_n, int splits, int flavour,
                           int mutual, float ratio, float dist)
{
    if (npairs <= 0 || max_n <= 0) return;
    hipLaunchKernelGGL(match_reduce_kernel, dim3((max_n + NT - 1) / NT, npairs, 2), dim3(NT), 0, st, fin_dev, splits);
    hipLaunchKernelGGL(match_decide_kernel, dim3((max_n + NT - 1) / NT, npairs), dim3(NT), 0, st, fin_dev, flavour,
                       mutual, ratio, dist);
}
