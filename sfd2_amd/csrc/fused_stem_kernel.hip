// Fused stem: norm_RGB + conv1a (3->64, 3x3) + BN + ReLU + conv1b (64->64, 3x3, stride 2) + BN + ReLU
// in one kernel (nets/extractor.py:104, nets/sfd2.py:268-270,314-316).  The 64-channel full-resolution
// tensor between the two convolutions (245 MB at 1600x1200, written and re-read once) never leaves
// the CU: each block computes the 9 x 65 conv1a pixels its 4 x 32 conv1b outputs need into LDS
// (MFMA, K = 48), then runs conv1b from there (MFMA, K = 9 x 64).  HBM traffic: image in,
// H/2 x W/2 x 64 out.
//
// Persistent blocks (one per CU, 156 KB of LDS): the nine conv1b filter taps (72 KB) are staged into
// LDS once per block and stay there, so phase 2 has no barriers and no per-tile filter traffic; the
// image patch of the block's NEXT tile is fetched into registers before phase 2 of the current one
// and lands in LDS afterwards, so its latency hides behind the MFMAs.  Barriers are raw
// s_barrier + lgkmcnt(0) (LDS visibility only): a __syncthreads() would also drain vmcnt, i.e. wait
// for the prefetch and for the previous tile's output stores.
// Measured at 1600x1200: streamed taps, one tile per block 128 us -> resident taps 113 us -> persistent
// + prefetch (this file) see profiles/.
#include "sfd2_internal.h"
#include <stdlib.h>
#include <stdio.h>

#define NT 512           // 8 waves: wave -> (output row = wave >> 1, 32-channel half = wave & 1)
#define F_TH 4           // conv1b output rows per tile
#define F_TW 32
#define F_RH 9           // conv1a rows needed: 2*4 + 1
#define F_RW 65
#define F_RP (F_RH * F_RW)        // 585 conv1a pixels
#define F_IH 11          // image rows needed
#define F_IW 68          // image cols needed (67) + 1 so the zero-weight kx = 3 slot reads valid bytes
#define F_IPT ((F_IH * F_IW + NT - 1) / NT)   // image pixels per thread (2)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ h4_t f_cvt4(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// -DSFD2_STEM_TRACE: cycle stamps of block 0's waves 0 and 7 at the section boundaries of its first tiles, printed by the
// launcher after a few launches (profiles/r02_stem_trace.txt).
#ifdef SFD2_STEM_TRACE
__device__ unsigned long long g_stem_trace[2][16][8];
#define ST_STAMP(k_)                                                                          \
    if (blockIdx.x == 0 && (wave == 0 || wave == 7) && lane == 0 && tcount < 16)              \
        g_stem_trace[wave == 7][tcount][k_] = __builtin_readcyclecounter();
#else
#define ST_STAMP(k_)
#endif

// a / d for the four normalisation constants (0.229, 0.224, 0.225, 255), bit-identical to the IEEE division: Markstein's
// q = RN(a * r), e = fma(-q, d, a), q' = fma(e, r, q) with r = RN(1 / d) is the correctly rounded quotient for every float
// with 1e-30 <= |a| <= 1e30 -- checked exhaustively over all 2^32 inputs per constant (tools/verify_const_div.c,
// profiles/r02_const_div_exhaustive.txt).  Three operations instead of the ~11 of the division routine (v_div_scale x 2,
// quarter-rate v_rcp, fma chain, v_div_fmas, v_div_fixup).  Anything outside [1e-20, 1e20] (zero, non-finite, tiny, huge)
// takes the routine, behind a real branch (the empty asm keeps hipcc from computing both and selecting).
__device__ __forceinline__ float f_div_const(float a, float d, float r)
{
    const float m = fabsf(a);
    if (__builtin_expect(!(m >= 1e-20f && m <= 1e20f), 0)) {
        asm volatile("" ::: "memory");
        return __fdiv_rn(a, d);
    }
    const float q = __fmul_rn(a, r);
    return __fmaf_rn(__fmaf_rn(-q, d, a), r, q);
}

// LDS-only barrier: every wave's LDS writes are visible to the block afterwards; global loads / stores stay in flight
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(NT, 2)   // one block per CU (156 KB of LDS): two waves per SIMD, 256 registers each
void fused_stem_kernel(const float *__restrict__ img, int H, int W, int normalise,
                       const half_t *__restrict__ w1 /*[2][3][64][8] conv1a A fragments*/,
                       const float *__restrict__ sc1, const float *__restrict__ sh1,
                       const half_t *__restrict__ w2 /*[9][64 oc][64 ic] conv1b*/,
                       const float *__restrict__ sc2, const float *__restrict__ sh2,
                       half_t *__restrict__ out /*[H2][W2][64]*/, int H2, int W2, int tiles_x, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *X1 = smem;                                         // [F_RP][128 B], 16-B slots swizzled with (rec >> 1) & 7
    unsigned char *Wt = X1 + ((F_RP * 128 + 1023) & ~1023);           // [9][64][128 B], same swizzle
    half_t *IM = reinterpret_cast<half_t *>(Wt + 9 * 8192);           // [F_IH][F_IW][4]
    float *SS = reinterpret_cast<float *>(IM + F_IH * F_IW * 4);      // sc1, sh1, sc2, sh2 (64 each)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const size_t plane = (size_t)H * W;

    // all nine conv1b filter taps -> Wt, once per block: tap t = 8 one-KB chunks (8 rows each), 1 per wave
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r = wave * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((r >> 1) & 7);
        __builtin_amdgcn_global_load_lds((gbl_void_t *)(w2 + ((size_t)t * 64 + r) * 64 + slot * 8),
                                         (lds_void_t *)(Wt + t * 8192 + wave * 1024), 16, 0, 0);
    }
    if (tid < 64) { SS[tid] = sc1[tid]; SS[64 + tid] = sh1[tid]; SS[128 + tid] = sc2[tid]; SS[192 + tid] = sh2[tid]; }
    h8_t a1[3];                        // conv1a filter fragments of this wave's 32-channel half (wave & 1)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
        a1[ky] = *reinterpret_cast<const h8_t *>(w1 + ((size_t)((wave & 1) * 3 + ky) * 64 + lane) * 8);

    // raw image values of this thread's F_IPT patch pixels (fp32 planes or uint8 HWC), fetched one tile ahead
    // raw values as loaded (float bits, or the uint8 byte): converted and normalised when they are written to LDS, so that
    // requesting them never waits
    unsigned int pr[F_IPT][3];
    unsigned pr_inside = 0;   // bit k: patch pixel k of this thread lies inside the image
    int fpy[F_IPT], fpx[F_IPT];   // this thread's patch pixels (the same for every tile)
#pragma unroll
    for (int k = 0; k < F_IPT; ++k) {
        const int p = tid + k * NT;
        fpy[k] = p / F_IW;
        fpx[k] = p - fpy[k] * F_IW;
    }
#define FETCH_IMG(tile_)                                                                                   \
    {                                                                                                      \
        const int ftx = (tile_) % tiles_x, fty = (tile_) / tiles_x;                                        \
        const int fy0 = 2 * (fty * F_TH) - 2, fx0 = 2 * (ftx * F_TW) - 2;                                  \
        pr_inside = 0;                                                                                     \
        _Pragma("unroll") for (int k = 0; k < F_IPT; ++k) {                                                \
            const int p = tid + k * NT;                                                                    \
            const int iy = fy0 + fpy[k], ix = fx0 + fpx[k];                                                \
            unsigned int r = 0u, g = 0u, b = 0u;                                                           \
            if (p < F_IH * F_IW && iy >= 0 && iy < H && ix >= 0 && ix < W) {                               \
                const size_t o = (size_t)iy * W + ix;                                                      \
                pr_inside |= 1u << k;                                                                      \
                if (normalise & 2) { /* uint8 HWC ingest (extract_localization.py:165-186) */              \
                    const unsigned char *u = reinterpret_cast<const unsigned char *>(img) + o * 3;         \
                    const int sw = (normalise & 4) ? 2 : 0; /* BGR -> RGB (:165) */                        \
                    r = u[sw]; g = u[1]; b = u[2 - sw];                                                    \
                } else {                                                                                   \
                    const unsigned int *iu = reinterpret_cast<const unsigned int *>(img);                  \
                    r = iu[o]; g = iu[plane + o]; b = iu[2 * plane + o];                                   \
                }                                                                                          \
            }                                                                                              \
            pr[k][0] = r; pr[k][1] = g; pr[k][2] = b;                                                      \
        }                                                                                                  \
    }
    // normalise (astype(float32) / 255. for uint8, then norm_RGB: one IEEE sub + one IEEE div) and store as fp16
#define STORE_IMG()                                                                                        \
    _Pragma("unroll") for (int k = 0; k < F_IPT; ++k) {                                                    \
        const int p = tid + k * NT;                                                                        \
        float r, g, b;                                                                                     \
        if (normalise & 2) { r = (float)pr[k][0]; g = (float)pr[k][1]; b = (float)pr[k][2]; }              \
        else { r = __uint_as_float(pr[k][0]); g = __uint_as_float(pr[k][1]); b = __uint_as_float(pr[k][2]); } \
        if (pr_inside & (1u << k)) {   /* zero padding stays exactly zero */                               \
            if (normalise & 2) {                                                                           \
                r = f_div_const(r, 255.0f, 1.0f / 255.0f); g = f_div_const(g, 255.0f, 1.0f / 255.0f);      \
                b = f_div_const(b, 255.0f, 1.0f / 255.0f);                                                 \
            }                                                                                              \
            if (normalise & 1) {                                                                           \
                r = f_div_const(__fsub_rn(r, 0.485f), 0.229f, 1.0f / 0.229f);                              \
                g = f_div_const(__fsub_rn(g, 0.456f), 0.224f, 1.0f / 0.224f);                              \
                b = f_div_const(__fsub_rn(b, 0.406f), 0.225f, 1.0f / 0.225f);                              \
            }                                                                                              \
        }                                                                                                  \
        if (p < F_IH * F_IW) *reinterpret_cast<h4_t *>(IM + p * 4) = f_cvt4(r, g, b, 0.0f);                \
    }

    // ---- phase 1 geometry, the same for every tile: a unit = (32-pixel column block, this wave's 32-channel half); 19
    // blocks over four wave pairs = 5, 5, 5, 4 units.  Per unit: image-patch read offset, X1 record offset and swizzle,
    // region coordinates for the image-bounds test (a measured tile spent ~1 150 cycles per unit around 96 cycles of MFMA,
    // most of it this arithmetic recomputed per tile: profiles/r02_stem_trace.txt).
    constexpr int P1_UNITS = ((F_RP + 31) / 32 + 3) / 4;
    const int ct1 = wave & 1;
    const int p1_n = ((F_RP + 31) / 32 - (wave >> 1) + 3) / 4;
    int p1_im[P1_UNITS], p1_x[P1_UNITS], p1_sw[P1_UNITS], p1_ry[P1_UNITS], p1_rx[P1_UNITS];
    bool p1_ok[P1_UNITS];
#pragma unroll
    for (int i = 0; i < P1_UNITS; ++i) {
        const int p = ((wave >> 1) + 4 * i) * 32 + lrow;
        const int pc = p < F_RP ? p : F_RP - 1;
        const int ry = pc / F_RW, rx = pc - ry * F_RW;
        p1_ok[i] = p < F_RP;
        p1_ry[i] = ry;
        p1_rx[i] = rx;
        p1_im[i] = (ry * F_IW + rx + 2 * lhi) * 4;                       // halfs
        p1_x[i] = (p ^ ((p >> 4) & 1)) * 128 + 8 * lhi;                  // bytes; + ((channel slot ^ swizzle) << 4)
        p1_sw[i] = ((p >> 1) & 7) << 4;
    }
    float4 s1[4], h1[4];                                   // conv1a scale / shift of this wave's channels
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s1[q] = *reinterpret_cast<const float4 *>(sc1 + ct1 * 32 + 8 * q + 4 * lhi);
        h1[q] = *reinterpret_cast<const float4 *>(sh1 + ct1 * 32 + 8 * q + 4 * lhi);
    }

    int tile = blockIdx.x;
    FETCH_IMG(tile)
    STORE_IMG()
    if (tile + (int)gridDim.x < n_tiles) FETCH_IMG(tile + (int)gridDim.x)   // the second tile's patch: written to LDS a tile later
    SFD2_BARRIER_DRAIN();   // full barrier once: the filter copies (vmcnt) and IM / SS (LDS) are complete

    int tcount = 0;
    (void)tcount;
    for (;; ++tcount) {
        ST_STAMP(0)
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int oy0 = ty * F_TH, ox0 = tx * F_TW;
        const int ry0 = 2 * oy0 - 1, rx0 = 2 * ox0 - 1;               // image coords of conv1a region pixel (0, 0)

        // ---- phase 1: conv1a on the 585 region pixels, 32 per MFMA column block.  A unit of work is (column block, 32-channel
        // half): 38 units over 8 waves = at most 5 per wave (whole blocks were 3 + 3 + 3 + 2 + ... = 6 half-units on the
        // busiest waves); a wave's half is fixed (wave & 1), so it keeps three filter fragments instead of six.  (Measured: 85.0 ->
        // 84.4 us -- the phases are serialised VALU / LDS / MFMA sections, not this imbalance.)
#pragma unroll
        for (int i = 0; i < P1_UNITS; ++i) {
            if (i < p1_n) {                                // wave-uniform: the last wave pair has one unit less
                f32x16_t acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const half_t *src = IM + p1_im[i] + ky * (F_IW * 4);
                    const h4_t lo = *reinterpret_cast<const h4_t *>(src);
                    const h4_t hi = *reinterpret_cast<const h4_t *>(src + 4);
                    h8_t b;
                    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
                    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ky], b, acc, 0, 0, 0);
                }
                // conv1b zero-pads conv1a's OUTPUT: region pixels outside the image are zeros, not conv1a(0); the select is
                // on the packed result (no branch around the epilogue)
                const int gy = ry0 + p1_ry[i], gx = rx0 + p1_rx[i];
                const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h4_t v = f_cvt4(fmaxf(acc[4 * q + 0] * s1[q].x + h1[q].x, 0.0f), fmaxf(acc[4 * q + 1] * s1[q].y + h1[q].y, 0.0f),
                                    fmaxf(acc[4 * q + 2] * s1[q].z + h1[q].z, 0.0f), fmaxf(acc[4 * q + 3] * s1[q].w + h1[q].w, 0.0f));
                    uint2 pk;
                    __builtin_memcpy(&pk, &v, 8);
                    if (!inside) pk = make_uint2(0u, 0u);
                    // records are stored pair-swapped where bit 4 of the index is set: the stride-2 reads of phase 2
                    // (lanes 256 B apart) then alternate between the two 128-byte halves of the bank space
                    if (p1_ok[i]) *reinterpret_cast<uint2 *>(X1 + p1_x[i] + ((((ct1 * 4 + q) << 4)) ^ p1_sw[i])) = pk;
                }
            }
        }
        ST_STAMP(1)
        const int next = tile + (int)gridDim.x, next2 = next + (int)gridDim.x;
        const bool has_next = next < n_tiles;
        ST_STAMP(2)
        LDS_BARRIER();                     // X1 complete; IM is free from here on
        ST_STAMP(3)

        // ---- phase 2: conv1b (stride 2) from X1 and the resident taps: wave -> (output row, 32-channel half)
        const int orow = wave >> 1, cth = wave & 1;
        f32x16_t acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.0f;
        const int ar = cth * 32 + lrow;
        const int a_off = ar * 128, a_sw = (ar >> 1) & 7;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int q = (2 * orow + ky) * F_RW + 2 * lrow + kx;
            const unsigned char *xq = X1 + (q ^ ((q >> 4) & 1)) * 128;
            const int bsw = (q >> 1) & 7;
            const unsigned char *wt = Wt + tap * 8192;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int slot = kk * 2 + lhi;
                const h8_t b = *reinterpret_cast<const h8_t *>(xq + ((slot ^ bsw) << 4));
                const h8_t a = *reinterpret_cast<const h8_t *>(wt + a_off + ((slot ^ a_sw) << 4));
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
            }
        }

        ST_STAMP(4)
        // the NEXT tile's patch (requested a whole tile ago) -> IM, in front of this tile's output stores: the wait hipcc
        // places in front of it then finds nothing younger than that request in flight (behind the stores it waited for
        // them: ~1 000 cycles of every tile), and the registers are free for the request of the tile after next
        if (has_next) STORE_IMG()
        const int oy = oy0 + orow, ox = ox0 + lrow;
        const bool inb = oy < H2 && ox < W2;
        half_t *o = out + ((size_t)(inb ? oy : 0) * W2 + (inb ? ox : 0)) * 64;
        // lanes l and l+32 hold the two 8-byte halves of a 16-byte channel run: regroup two quads with
        // v_permlane32_swap so every lane issues one 16-byte store per quad pair (as conv_igemm2)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            uint2 pk[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 2 * m + j;
                const int c0 = cth * 32 + 8 * q + 4 * lhi;
                const float4 s = *reinterpret_cast<const float4 *>(SS + 128 + c0);
                const float4 h = *reinterpret_cast<const float4 *>(SS + 192 + c0);
                const h4_t hv = f_cvt4(fmaxf(acc2[4 * q + 0] * s.x + h.x, 0.0f), fmaxf(acc2[4 * q + 1] * s.y + h.y, 0.0f),
                                       fmaxf(acc2[4 * q + 2] * s.z + h.z, 0.0f), fmaxf(acc2[4 * q + 3] * s.w + h.w, 0.0f));
                __builtin_memcpy(&pk[j], &hv, 8);
            }
            const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
            if (inb) *reinterpret_cast<uint4 *>(o + cth * 32 + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
        }
        ST_STAMP(5)
        if (!has_next) break;
        if (next2 < n_tiles) FETCH_IMG(next2)   // global loads only; consumed after phase 2 of the next tile
        ST_STAMP(6)
        LDS_BARRIER();                     // IM complete, and every wave is done reading X1
        ST_STAMP(7)
        tile = next;
    }
#undef FETCH_IMG
#undef STORE_IMG
}

void launch_fused_stem(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *w1, const float *sc1,
                       const float *sh1, const half_t *w2, const float *sc2, const float *sh2, half_t *out, int H2, int W2)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = (size_t)((F_RP * 128 + 1023) & ~1023) + 9 * 8192 + (size_t)F_IH * F_IW * 4 * sizeof(half_t) + 256 * sizeof(float);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stem_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;   // 156 KB of LDS: one resident block per CU
        attr_done = true;
    }
    const int tiles_x = (W2 + F_TW - 1) / F_TW, tiles_y = (H2 + F_TH - 1) / F_TH;
    const int n_tiles = tiles_x * tiles_y;
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    hipLaunchKernelGGL(fused_stem_kernel, dim3(grid), dim3(NT), lds, st, img, H, W, normalise, w1, sc1, sh1, w2, sc2, sh2, out,
                       H2, W2, tiles_x, n_tiles);
#ifdef SFD2_STEM_TRACE
    {
        static int dumps = 0;
        if (H >= 1000 && ++dumps == 40) {
            (void)hipStreamSynchronize(st);
            static unsigned long long h[2][16][8];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stem_trace), sizeof(h));
            fprintf(stderr, "stemtrace columns: phase 1 | - | barrier wait | phase 2 MFMAs | image -> LDS + epilogue + stores | fetch issue (tile + 2) | barrier wait\n");
            for (int w = 0; w < 2; ++w)
                for (int t = 2; t < 10; ++t) {
                    fprintf(stderr, "stemtrace wave %d tile %2d:", w * 7, t);
                    for (int k = 1; k < 8; ++k) fprintf(stderr, " %6lld", (long long)(h[w][t][k] - h[w][t][k - 1]));
                    fprintf(stderr, "  (tile %lld)\n", (long long)(h[w][t + 1][0] - h[w][t][0]));
                }
        }
    }
#endif
}
