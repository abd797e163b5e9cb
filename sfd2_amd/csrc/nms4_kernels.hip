// simple_nms (nets/extractor.py:20-35) for the radius the pipeline uses (4) + threshold + border + compaction
// (nets/extractor.py:158-183), one kernel.  Own translation unit because it is compiled with -fno-honor-nans
// (sfd2_amd/build.py): the kernel is VALU-issue bound and without the flag hipcc puts a NaN-canonicalising v_max x, x in
// front of the fmaxf operands it cannot prove quiet (32 instead of 22 v_max per eight pooled values).  Heat maps are
// products of a soft-max probability and a stability weight; a NaN score has no defined rank in the reference either.
#include "sfd2_internal.h"
#include <math.h>

// Same algorithm, restructured for the LDS: region 64 x 128 (tile 24 x 88 + 20-px halo), only two
// float planes (scores S, row-pass result A).  Each max-pool is a register-blocked row pass
// (8 outputs from 16 loaded values: suffix/prefix maxima) and a column pass whose result is
// compared in registers and turned into bit masks with wave ballots; the two mask dilations
// (supp_mask = max_pool(max_mask) > 0) are bit operations on 128-bit rows.  The suppressed score
// map (where(supp, 0, scores)) is formed on the fly from S and the supp bits.
#define N2_TW 88
#define N2_HALO 20
#define N2_RW 128
#define N2_SP 136   // S row pitch: 4 pad floats (-inf) on either side
// region rows RH (64 or 136) is a template parameter: tile rows = RH - 2 * halo, A plane rows = RH + 8

__device__ __forceinline__ void pool8(const float (&v)[16], float (&o)[8])
{
    // o[j] = max(v[j .. j+8]) = max(suffix max of v[0..7] at j, prefix max of v[8..15] at j)
    float suf[8], pre[8];
    suf[7] = v[7];
#pragma unroll
    for (int j = 6; j >= 0; --j) suf[j] = fmaxf(v[j], suf[j + 1]);
    pre[0] = v[8];
#pragma unroll
    for (int j = 1; j < 8; ++j) pre[j] = fmaxf(pre[j - 1], v[8 + j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = fmaxf(suf[j], pre[j]);
}

// Row pass.  MASKED: where(supp, 0, scores) is formed from the supp bits (already restricted to pixels inside the image,
// so padding stays -inf) and ALSO written back into S for the eight centre columns: supp only grows from one round to
// the next, so round 2 and the column passes read the suppressed map directly.  Neighbouring threads read those
// columns as part of their 16-wide windows while they are being rewritten; every reader applies the same bits itself,
// so it sees the same values either way.
//
// Only the tile has to be exact at the end, and every round consumes 8 rows / columns of halo (4 for the dilation, 4 for the
// pool): round K's row pass covers rows [8K, RH - 8K) and the 8-column units [K, 16 - K), its column pass the 8-row
// blocks [K, RH / 8 - K).  Words and values outside those ranges go stale; nothing inside the next round's range reads them.
template <bool MASKED, int N2_RH, int K>
__device__ __forceinline__ void n2_row_pass(float *__restrict__ S, float *__restrict__ A,
                                            const unsigned long long *__restrict__ supp)
{
    constexpr int Y0 = 8 * K, NY = N2_RH - 16 * K, U0 = K, NU = 16 - 2 * K;
    for (int i = threadIdx.x; i < NY * NU; i += blockDim.x) {
        const int yy = i / NU;
        const int y = Y0 + yy, x0 = (U0 + i - yy * NU) * 8;
        float v[16];
        const float4 *src = reinterpret_cast<const float4 *>(S + y * N2_SP + x0);   // region cols x0-4 .. x0+11
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = src[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
        if (MASKED) {
            // bits of region cols x0-4 .. x0+11, taken from the row mask shifted left by 4
            const unsigned long long w0 = supp[2 * y], w1 = supp[2 * y + 1];
            const unsigned long long lo = w0 << 4, hi = (w1 << 4) | (w0 >> 60), top = w1 >> 60;
            unsigned long long bits;
            if (x0 < 64) bits = (lo >> x0) | (x0 ? (hi << (64 - x0)) : 0ull);
            else { const int sft = x0 - 64; bits = (hi >> sft) | (sft ? (top << (64 - sft)) : 0ull); }
            const int keep = (int)~(unsigned int)bits;
#pragma unroll
            for (int i = 0; i < 16; ++i) {   // one v_bfe_i32 (bit i -> all ones / zero) + one v_and per value; written
                int t;                       // in asm because hipcc turns the builtin into extract + compare + select
                asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(t) : "v"(keep), "n"(i));
                v[i] = __uint_as_float(__float_as_uint(v[i]) & (unsigned int)t);
            }
            float4 *ctr = reinterpret_cast<float4 *>(S + y * N2_SP + x0 + 4);
            ctr[0] = make_float4(v[4], v[5], v[6], v[7]);
            ctr[1] = make_float4(v[8], v[9], v[10], v[11]);
        }
        float o[8];
        pool8(v, o);
        float4 *dst = reinterpret_cast<float4 *>(A + (y + 4) * N2_RW + x0);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// Column pass + compare: eq[y] = ballot(s == max_pool(s)) on the (suppressed) score map, 64 columns per word.  Which of
// these bits count (inside the image, not suppressed) is settled on whole words by n2_merge / n2_dilate.
template <int N2_RH, int K>
__device__ __forceinline__ void n2_col_pass(const float *__restrict__ S, const float *__restrict__ A,
                                            unsigned long long *__restrict__ eq)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int u = wave; u < (N2_RH / 8 - 2 * K) * 2; u += nw) {
        const int k = K + (u >> 1), h = u & 1, x = h * 64 + lane;
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = A[(8 * k + i) * N2_RW + x];
        float o[8];
        pool8(a, o);
        // lane j keeps row 8k + j's word: one store per block of rows.  (v_writelane_b32 from the ballot's SGPRs would halve
        // these selects; written in asm behind v_readfirstlane it produced wrong masks -- a VALU-writes-SGPR hazard that
        // hipcc does not pad inside asm statements -- and with the s_nops it saves nothing.)
        int mlo = 0, mhi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sv = S[(8 * k + j) * N2_SP + 4 + x];
            const unsigned long long word = __ballot(sv == o[j]);
            if (lane == j) { mlo = (int)(unsigned int)word; mhi = (int)(unsigned int)(word >> 32); }
        }
        if (lane < 8) eq[2 * (8 * k + lane) + h] = ((unsigned long long)(unsigned int)mhi << 32) | (unsigned int)mlo;
    }
}

// max_mask after a round, on whole words: FIRST: mask = eq & valid; else mask |= eq & valid & ~supp
template <bool FIRST>
__device__ __forceinline__ unsigned long long n2_merged(const unsigned long long *__restrict__ m, const unsigned long long *__restrict__ eq,
                                                        const unsigned long long *__restrict__ valid,
                                                        const unsigned long long *__restrict__ supp, int i)
{
    const unsigned long long e = eq[i] & valid[i];
    return FIRST ? e : (m[i] | (e & ~supp[i]));
}

// folds the last column pass into max_mask, then supp_mask = max_pool(max_mask) > 0 restricted to the image
template <int N2_RH, bool FIRST>
__device__ __forceinline__ void n2_dilate(unsigned long long *__restrict__ m, unsigned long long *__restrict__ tmp,
                                          unsigned long long *__restrict__ supp, const unsigned long long *__restrict__ eq,
                                          const unsigned long long *__restrict__ valid)
{
    for (int u = threadIdx.x; u < N2_RH; u += blockDim.x) {
        const unsigned long long w0 = n2_merged<FIRST>(m, eq, valid, supp, 2 * u), w1 = n2_merged<FIRST>(m, eq, valid, supp, 2 * u + 1);
        m[2 * u] = w0;
        m[2 * u + 1] = w1;
        unsigned long long d0 = w0, d1 = w1;
#pragma unroll
        for (int sft = 1; sft <= 4; ++sft) {
            d0 |= (w0 << sft) | (w0 >> sft) | (w1 << (64 - sft));
            d1 |= (w1 << sft) | (w1 >> sft) | (w0 >> (64 - sft));
        }
        tmp[2 * u] = d0;
        tmp[2 * u + 1] = d1;
    }
    __syncthreads();
    for (int u = threadIdx.x; u < N2_RH * 2; u += blockDim.x) {
        const int y = u >> 1, h = u & 1;
        unsigned long long d = 0ull;
#pragma unroll
        for (int dy = -4; dy <= 4; ++dy)
            if (y + dy >= 0 && y + dy < N2_RH) d |= tmp[2 * (y + dy) + h];
        supp[u] = d & valid[u];
    }
    __syncthreads();
}

template <int N2_RH>
__global__ __launch_bounds__(1024)
void nms4_select_kernel(const float *__restrict__ heat, int H, int W, float conf_th, int border, int Hb, int Wb,
                        float *__restrict__ nms_dense, unsigned long long *__restrict__ cand, int cand_cap,
                        unsigned int *__restrict__ counters, unsigned int *__restrict__ hist, int fuse_threshold, int top_k)
{
    constexpr int N2_TH = N2_RH - 2 * N2_HALO, N2_AR = N2_RH + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *S = reinterpret_cast<float *>(smem);                         // [RH][SP]
    float *A = S + N2_RH * N2_SP;                                       // [RH + 8][RW]
    unsigned long long *M = reinterpret_cast<unsigned long long *>(A + N2_AR * N2_RW);   // [RH][2] max_mask
    unsigned long long *SU = M + 2 * N2_RH;                             // supp_mask
    unsigned long long *TM = SU + 2 * N2_RH;                            // scratch of the dilation
    unsigned long long *VL = TM + 2 * N2_RH;                            // pixel lies inside the image
    unsigned long long *EQ = VL + 2 * N2_RH;                            // last column pass: s == max_pool(s)
    const float NEG = -INFINITY;
    const int gy0 = blockIdx.y * N2_TH - N2_HALO, gx0 = blockIdx.x * N2_TW - N2_HALO;

    // region -> S in groups of four columns (34 per row: the region's 32 and the two -inf pads); gx0 is a multiple of 4, so
    // with W % 4 == 0 a group is one aligned 16-byte load
    const bool w4 = (W & 3) == 0;
    for (int i = threadIdx.x; i < N2_RH * (N2_SP / 4); i += blockDim.x) {
        const int y = i / (N2_SP / 4), g = i - y * (N2_SP / 4);
        const int gy = gy0 + y, gx = gx0 + 4 * g - 4;
        float4 v = make_float4(NEG, NEG, NEG, NEG);
        if (g >= 1 && g <= N2_RW / 4 && gy >= 0 && gy < H && gx + 3 >= 0 && gx < W) {
            const float *row = heat + (size_t)gy * W;
            if (w4 && gx >= 0 && gx + 3 < W) {
                v = *reinterpret_cast<const float4 *>(row + gx);
            } else {
                if (gx >= 0) v.x = row[gx];
                if (gx + 1 >= 0 && gx + 1 < W) v.y = row[gx + 1];
                if (gx + 2 >= 0 && gx + 2 < W) v.z = row[gx + 2];
                if (gx + 3 < W) v.w = row[gx + 3];
            }
        }
        *reinterpret_cast<float4 *>(S + y * N2_SP + 4 * g) = v;
    }
    for (int i = threadIdx.x; i < 4 * N2_RW; i += blockDim.x) {          // -inf rows above / below A
        A[i] = NEG;
        A[(N2_RH + 4) * N2_RW + i] = NEG;
    }
    for (int u = threadIdx.x; u < N2_RH * 2; u += blockDim.x) {         // columns [c0, c1) of row y are inside the image
        const int y = u >> 1, h = u & 1, gy = gy0 + y;
        int c0 = -gx0 - 64 * h, c1 = W - gx0 - 64 * h;
        c0 = c0 < 0 ? 0 : (c0 > 64 ? 64 : c0);
        c1 = c1 < 0 ? 0 : (c1 > 64 ? 64 : c1);
        unsigned long long w = 0ull;
        if (gy >= 0 && gy < H && c1 > c0) w = (c1 == 64 ? ~0ull : ((1ull << c1) - 1ull)) & ~((1ull << c0) - 1ull);
        VL[u] = w;
    }
    __syncthreads();
    n2_row_pass<false, N2_RH, 0>(S, A, nullptr);
    __syncthreads();
    n2_col_pass<N2_RH, 0>(S, A, EQ);                                            // max_mask = scores == max_pool(scores)
    __syncthreads();
    n2_dilate<N2_RH, true>(M, TM, SU, EQ, VL);                                  // supp_mask = max_pool(max_mask) > 0
    n2_row_pass<true, N2_RH, 1>(S, A, SU);
    __syncthreads();
    n2_col_pass<N2_RH, 1>(S, A, EQ);
    __syncthreads();
    n2_dilate<N2_RH, false>(M, TM, SU, EQ, VL);                                 // max_mask |= new_max_mask & ~supp_mask; next supp
    n2_row_pass<true, N2_RH, 2>(S, A, SU);
    __syncthreads();
    n2_col_pass<N2_RH, 2>(S, A, EQ);
    __syncthreads();
    for (int u = threadIdx.x; u < N2_RH * 2; u += blockDim.x) M[u] = n2_merged<false>(M, EQ, VL, SU, u);
    // Candidates are first gathered per block in LDS (the A plane is free now) so that the global
    // cursor sees ONE atomic per block: tens of thousands of same-address atomics (~12 ns each at
    // the L2) were the whole cost of this kernel.
    unsigned long long *lkeys = reinterpret_cast<unsigned long long *>(A);   // <= TH * 88 keys (16.5 / 66 KB), A holds AR * 128 floats
    // (all LDS stays in the one dynamic array: a static __shared__ would shift its 16-byte base)
    unsigned int &l_cnt = reinterpret_cast<unsigned int *>(EQ + 2 * N2_RH)[0];
    unsigned int &l_base = reinterpret_cast<unsigned int *>(EQ + 2 * N2_RH)[1];
    if (threadIdx.x == 0) l_cnt = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N2_TH * N2_TW; i0 += blockDim.x) {             // wave-uniform trip count (ballots below)
        const int i = i0 + threadIdx.x;
        const int ty = i / N2_TW, tx = i - ty * N2_TW;
        const int gy = blockIdx.y * N2_TH + ty, gx = blockIdx.x * N2_TW + tx;
        bool is_cand = false;
        unsigned long long key = 0ull;
        if (i < N2_TH * N2_TW && gy < H && gx < W) {
            const int ry = ty + N2_HALO, rx = tx + N2_HALO;
            const bool mk = (M[2 * ry + (rx >> 6)] >> (rx & 63)) & 1ull;
            const float v = mk ? heat[(size_t)gy * W + gx] : 0.0f;       // where(max_mask, scores, zeros); S holds the suppressed map by now
            if (nms_dense) nms_dense[(size_t)gy * W + gx] = v;
            if (cand && v > conf_th && gx >= border && gx < Wb - border && gy >= border && gy < Hb - border) {
                const unsigned int idx = (unsigned int)(gy * W + gx);
                key = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
                is_cand = true;
            }
        }
        // one LDS atomic per wave (a few hundred same-address atomics per block serialise otherwise)
        const unsigned long long mc = __ballot(is_cand);
        if (mc) {
            const int lane = threadIdx.x & 63;
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&l_cnt, (unsigned int)__popcll(mc));
            base = __shfl(base, 0);
            if (is_cand) lkeys[base + (unsigned int)__popcll(mc & ((1ull << lane) - 1ull))] = key;
        }
    }
    __syncthreads();
    if (!cand) return;
    if (l_cnt != 0) {
        if (threadIdx.x == 0) l_base = atomicAdd(&counters[0], l_cnt);
        __syncthreads();
        for (unsigned int i = threadIdx.x; i < l_cnt; i += blockDim.x) {
            const unsigned int pos = l_base + i;
            if (pos < (unsigned int)cand_cap) {
                const unsigned long long key = lkeys[i];
                cand[pos] = key;
                atomicAdd(&hist[key_bin(key)], 1u);
            }
        }
    }
    if (!fuse_threshold) return;
    // the last block to finish (ticket in counters[7]) searches the top-K threshold in the now complete histogram: the
    // selection chain behind this kernel starts at the compaction (one launch less).  Histogram and count are atomics, read
    // back with device-scope atomic loads: no fence (sfd2_select_threshold).
    SFD2_BARRIER_DRAIN();   // this block's atomics are acknowledged: the vmcnt(0) is written out -- a workgroup-scope
                            // __syncthreads() does not wait for global operations on this target
    if (threadIdx.x == 0) l_base = atomicAdd(&counters[7], 1u) == gridDim.x * gridDim.y - 1 ? 1u : 0u;
    __syncthreads();
    if (!l_base) return;
    sfd2_select_threshold<true>(cand_cap, top_k, counters, reinterpret_cast<unsigned int *>(lkeys));
}

void launch_nms4_select(hipStream_t st, const float *heat, int H, int W, float conf_th, int border, int Hb, int Wb, float *nms_dense,
                        unsigned long long *cand, int cand_cap, unsigned int *counters, unsigned int *hist, int fuse_threshold, int top_k)
{
    // The kernel is VALU-issue bound (profiles/r02_nms_pmc.txt: 1 215 VALU instructions per thread, 8 waves per SIMD),
    // so what counts is the number of REGION pixels the busiest CU has to process.  Two tile heights: 24 rows (region
    // 64 x 128, 73 KB of LDS, two blocks per CU, 3.9x halo overhead) or 96 rows (region 136 x 128, 155 KB, one block
    // per CU, 2.1x) -- whichever gives the busiest CU less to do for this image (1600x1200: 247 big tiles = one
    // round on 256 CUs instead of 950 small ones).
    static bool attr4 = false;
    auto lds_of = [](int rh) { return (size_t)(rh * N2_SP + (rh + 8) * N2_RW) * sizeof(float) + 5 * 2 * rh * sizeof(unsigned long long) + 16; };
    if (!attr4) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(nms4_select_kernel<64>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(nms4_select_kernel<136>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(136));
        attr4 = true;
    }
    const int gx = (W + N2_TW - 1) / N2_TW;
    const long long nb_s = (long long)gx * ((H + 23) / 24), nb_b = (long long)gx * ((H + 95) / 96);
    const long long work_s = ((nb_s + 511) / 512) * 2 * 64, work_b = ((nb_b + 255) / 256) * 136;   // region rows on the busiest CU
    static const char *force = sfd2_env("SFD2_NMS_TILE");
    const bool big = force ? force[0] == 'b' : work_b < work_s;
    if (big) hipLaunchKernelGGL(nms4_select_kernel<136>, dim3(gx, (H + 95) / 96), dim3(1024), lds_of(136), st, heat, H, W, conf_th,
                                border, Hb, Wb, nms_dense, cand, cand_cap, counters, hist, fuse_threshold, top_k);
    else hipLaunchKernelGGL(nms4_select_kernel<64>, dim3(gx, (H + 23) / 24), dim3(1024), lds_of(64), st, heat, H, W, conf_th,
                            border, Hb, Wb, nms_dense, cand, cand_cap, counters, hist, fuse_threshold, top_k);
}
