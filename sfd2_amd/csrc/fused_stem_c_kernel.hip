// Fused stem, compensated (SFD2_PREC_F16C): norm_RGB + conv1a (3->64) + BN + ReLU + conv1b (64->64, stride 2) + BN + ReLU
// in one kernel, with conv1a's output kept in LDS as a hi plane AND a corr plane (sfd2_internal.h) and conv1b
// compensated like every other backbone layer.  nets/extractor.py:104, nets/sfd2.py:268-270,314-316.
//
// Same tiling as fused_stem_kernel (a block computes the 9 x 65 conv1a pixels its 4 x 32 conv1b outputs need), but the
// two planes of that region are 146 KB of the CU's 160 KB, so the conv1b filters (147 KB for hi + corr) cannot live in
// LDS beside them.  They live in REGISTERS instead, which only works if a wave needs few of them: the K dimension of
// conv1b (9 taps x 2 halves of 32 input channels = 18 units) is split over four wave groups (5 + 5 + 4 + 4 units; a unit
// = 8 + 8 registers of hi and corr fragments), each wave accumulating a PARTIAL sum of all four output rows for its
// 32-channel half.  The partials of the three rows a wave does not finish go through LDS (over the dead conv1a region)
// and are added in a fixed order -- wave group 0, 1, 2, 3 -- so results do not depend on timing.
//
//   phase 1  conv1a on the 585 region pixels: image = hi + lo fp16, filters = hi + lo fp16, three fp16 MFMA passes;
//            epilogue -> X1h (fp16) and X1c (corr units)                                   [barrier]
//   phase 2  conv1b partial sums: per unit and output row two fp16 MFMAs on X1h and one fp8 MFMA on X1c  [barrier]
//   phase 3  partials -> LDS                                                                [barrier]
//   phase 4  row (wave >> 1) of this wave's channel half: sum of the four partials, BN + ReLU, hi + corr planes out;
//            the next tile's image patch (fetched into registers a tile ago) -> LDS         [barrier]
// Persistent blocks, one per CU.  HBM traffic: image in, H/2 x W/2 x 64 x 4 B out (the unfused compensated pair moves
// 1 GB through HBM for the full-resolution planes: 650 us at 1600x1200).
#include "sfd2_internal.h"
#include <stdlib.h>

#define SC_NT 512
#define SC_TH 4
#define SC_TW 32
#define SC_RH 9
#define SC_RW 65
#define SC_RP (SC_RH * SC_RW)          // 585 conv1a pixels
#define SC_IH 11
#define SC_IW 68
#define SC_IPT ((SC_IH * SC_IW + SC_NT - 1) / SC_NT)
#define SC_X1 (SC_RP * 128)            // bytes of one plane of the region
#define SC_IM (SC_IH * SC_IW * 8)      // bytes of one image plane ([px][4] fp16)
#define SC_LDS (2 * SC_X1 + 2 * SC_IM + 1024)
#define SC_NU 5                        // units per wave (the last two wave groups run four)
// slot term of a region pixel's record.  (pixel >> 1) & 7 is what phase 2's stride-2 fragment reads need (sixteen lanes = sixteen pixels two apart); the
// parity bit on top changes nothing for them (a read's pixels share their parity: one constant XOR) and separates pixels 2 m and 2 m + 1 in
// phase 1's 16-byte corr stores, whose eight-lane groups are eight CONSECUTIVE pixels: two-way conflicted until the end of round 5
// (-DSFD2_STEMC_OLDSWZ).  The 8-byte hi stores stay two-way (CHANGELOG round 5).
#ifdef SFD2_STEMC_OLDSWZ
#define SC_SWZ(q_) (((q_) >> 1) & 7)
#define SC_SWZ_PAR(c_) 0
#else
#define SC_SWZ(q_) ((((q_) >> 1) & 7) ^ (((q_) & 1) << 2))
#define SC_SWZ_PAR(c_) (((c_) & 1) << 2)      // the parity part alone, of a pixel index known up to an even offset
#endif
#ifndef SFD2_STEMC_P1MODE
#define SFD2_STEMC_P1MODE 0
#endif

__device__ __forceinline__ float sc_div_const(float a, float d, float r)   // see fused_stem_kernel.hip (exact for these constants)
{
    const float m = fabsf(a);
    if (__builtin_expect(!(m >= 1e-20f && m <= 1e20f), 0)) {
        asm volatile("" ::: "memory");
        return __fdiv_rn(a, d);
    }
    const float q = __fmul_rn(a, r);
    return __fmaf_rn(__fmaf_rn(-q, d, a), r, q);
}

#define SC_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// -DSFD2_STEMC_TRACE: cycle stamps of all eight waves of one block at the section boundaries of its first tiles, printed by the
// launcher after a few launches
#ifdef SFD2_STEMC_TRACE
#include <stdio.h>
__device__ unsigned long long g_stemc_trace[8][16][12];
#define SC_STAMP(k_)                                                                          \
    if (blockIdx.x == 3 && lane == 0 && tcount < 16)                                          \
        g_stemc_trace[wave][tcount][k_] = __builtin_readcyclecounter();
#else
#define SC_STAMP(k_)
#endif

// X3 epilogue of four consecutive channels: y = relu(a * scale + shift) as hi = fp16(y) and lo' = fp16((y - hi) * 2^11)
// (x3_split's arithmetic, conv_f32_kernels.hip)
__device__ __forceinline__ void sc_x3_epi4(float a0, float a1, float a2, float a3, float4 sc, float4 sh, uint2 &hv, uint2 &lv)
{
    const float v0 = fmaxf(a0 * sc.x + sh.x, 0.0f), v1 = fmaxf(a1 * sc.y + sh.y, 0.0f);
    const float v2 = fmaxf(a2 * sc.z + sh.z, 0.0f), v3 = fmaxf(a3 * sc.w + sh.w, 0.0f);
    const h4_t h = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
    const h4_t l = {(half_t)((v0 - (float)h[0]) * 2048.0f), (half_t)((v1 - (float)h[1]) * 2048.0f),
                    (half_t)((v2 - (float)h[2]) * 2048.0f), (half_t)((v3 - (float)h[3]) * 2048.0f)};
    __builtin_memcpy(&hv, &h, 8);
    __builtin_memcpy(&lv, &l, 8);
}

// sfd2_epi16_fp6's arithmetic (same operations in the same order: identical bits) cut into eight pieces of ~10 vector instructions with a slot
// between them: the caller's slot(k), k = 0..8, issues one MFMA of the NEXT unit's chain; sched_barrier(0) pins the interleave (hipcc otherwise
// keeps the chain in one run, and a wave stalls at its second MFMA until the pipe is free: the chain's 288 cycles are then exposed)
template <class SLOT>
__device__ __forceinline__ void sc_epi16_fp6_piped(const f32x16_t &acc, const float4 (&sc)[4], const float4 (&sh)[4], uint2 (&hv)[4], uint4 &rec0, uint4 &rec1,
                                                   float &mx, bool counted, SLOT slot)
{
    f32x16v_t v, l;
    float m = 0.0f;
    slot(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2_t v01 = f32x2_t{acc[4 * q], acc[4 * q + 1]} * f32x2_t{sc[q].x, sc[q].y} + f32x2_t{sh[q].x, sh[q].y};
        f32x2_t v23 = f32x2_t{acc[4 * q + 2], acc[4 * q + 3]} * f32x2_t{sc[q].z, sc[q].w} + f32x2_t{sh[q].z, sh[q].w};
        m = sfd2_max3(sfd2_max3(m, v01[0], v01[1]), v23[0], v23[1]);
        v01[0] = __builtin_amdgcn_fmed3f(v01[0], 0.0f, SFD2_C_SAT); v01[1] = __builtin_amdgcn_fmed3f(v01[1], 0.0f, SFD2_C_SAT);
        v23[0] = __builtin_amdgcn_fmed3f(v23[0], 0.0f, SFD2_C_SAT); v23[1] = __builtin_amdgcn_fmed3f(v23[1], 0.0f, SFD2_C_SAT);
        __builtin_amdgcn_sched_barrier(0);
        slot(2 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
        const h2_t h01 = __builtin_convertvector(v01, h2_t), h23 = __builtin_convertvector(v23, h2_t);
        __builtin_memcpy(&hv[q].x, &h01, 4);
        __builtin_memcpy(&hv[q].y, &h23, 4);
        const f32x2_t l01 = (v01 - f32x2_t{(float)h01[0], (float)h01[1]}) * 2048.0f, l23 = (v23 - f32x2_t{(float)h23[0], (float)h23[1]}) * 2048.0f;
        v[4 * q] = v01[0]; v[4 * q + 1] = v01[1]; v[4 * q + 2] = v23[0]; v[4 * q + 3] = v23[1];
        l[4 * q] = l01[0]; l[4 * q + 1] = l01[1]; l[4 * q + 2] = l23[0]; l[4 * q + 3] = l23[1];
        __builtin_amdgcn_sched_barrier(0);
        slot(2 * q + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifndef SFD2_NO_RANGE
    mx = counted ? __builtin_fmaxf(mx, m) : mx;
#endif
    const unsigned int mb = __float_as_uint(__builtin_fminf(__builtin_fmaxf(m, 0.0f), SFD2_C_SAT));
    v6i_t d;
#if SFD2_PIX6_BF6
    int e8 = (int)(mb >> 23) - 4 + ((mb & 0x7fffffu) > 0x600000u ? 1 : 0);
    e8 = e8 < 1 ? 1 : e8;
    const float scale = __uint_as_float((unsigned int)e8 << 23);
    asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(l), "v"(v), "v"(scale));
#else
    int e8 = (int)(mb >> 23) - 2 + ((mb & 0x7fffffu) > 0x700000u ? 1 : 0);
    e8 = e8 < 1 ? 1 : e8;
    const float scale = __uint_as_float((unsigned int)e8 << 23);
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(l), "v"(v), "v"(scale));
#endif
    rec0 = make_uint4((unsigned)d[0], (unsigned)d[1], (unsigned)d[2], (unsigned)d[3]);
    rec1 = make_uint4((unsigned)d[4], (unsigned)d[5], (unsigned)e8 * 0x01010101u, 0u);
}

// X3 (SFD2_PREC_F16X3): the same kernel with conv1a's output and conv1b's filters / output as hi / lo' fp16 pairs instead of hi / corr
// units: X1c holds lo', w2's second half the filters' lo' fragments, phase 2 runs hi x hi over the wave's units, scales the partial
// sums by 2^11 (exactly) and adds the cross terms hi x lo' + lo' x hi to the same accumulators (2^-11 goes into phase 4), the
// output planes are hi / lo'.
// O6: conv1b's corr records leave as fp6 half-records (sfd2_epi16_fp6) for a consumer that reads them with the fp6 x fp6 scaled MFMA --
// and conv1a's (LDS-resident) records are fp6 half-records too: phase 1 converts a lane's 16 channels with one
// v_cvt_scalef32_2xpk16_fp6_f32 instead of sixteen fp8 conversions, phase 2's correction is the 33.5-cycle fp6 x fp6 MFMA; w2's corr
// fragments are then fp6 strings in the half-records' channel order with the row's E8M0 scale byte in dword 6 (api_weights.hip).
template <bool X3, bool O6 = false>
__global__ __launch_bounds__(SC_NT, 2)
void fused_stem_c_kernel(const float *__restrict__ img, int H, int W, int normalise,
                         const half_t *__restrict__ w1 /*[2 hi/lo][2][3][64][8] conv1a A fragments*/,
                         const float *__restrict__ sc1, const float *__restrict__ sh1,
                         const unsigned char *__restrict__ w2 /*[2 cth][18 units][64 lanes][64 B]: hi K slices 0, 1, then the corr fragment*/,
                         const float *__restrict__ sc2, const float *__restrict__ sh2,
                         half_t *__restrict__ out, half_t *__restrict__ out_c /*[H2][W2][64] each*/, int H2, int W2,
                         int tiles_x, int n_tiles, int sa, unsigned int *__restrict__ range /* base of the context's range-status words, or null */)
{
    unsigned int smax1 = 0, smax2 = 0;     // range status of conv1a's (LDS-resident) and conv1b's output: wave-uniform across the tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *X1h = smem;                                   // [SC_RP][128 B], 16-B slots swizzled with SC_SWZ(rec), records pair-swapped
    unsigned char *X1c = smem + SC_X1;                           // the corr plane, same addressing
    half_t *IMh = reinterpret_cast<half_t *>(smem + 2 * SC_X1);  // [SC_IH][SC_IW][4]
    half_t *IMl = IMh + SC_IH * SC_IW * 4;
    float *SS = reinterpret_cast<float *>(smem + 2 * SC_X1 + 2 * SC_IM);   // sc1, sh1, sc2, sh2
    float *PT = reinterpret_cast<float *>(smem);                 // phase 3 / 4: partial tiles [cth][row][group][4 quads][64 lanes] float4, over X1

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const size_t plane = (size_t)H * W;
    const int cth = wave & 1, kg = wave >> 1;                    // channel half; K group (= the output row this wave finishes)
    const int u0 = kg < 2 ? kg * 5 : 10 + (kg - 2) * 4;
    const int nu = kg < 2 ? 5 : 4;

    if (tid < 64) { SS[tid] = sc1[tid]; SS[64 + tid] = sh1[tid]; SS[128 + tid] = sc2[tid]; SS[192 + tid] = sh2[tid]; }

    // ---- conv1b filter fragments of this wave's units, resident
    h8_t wa[SC_NU][2];
    v8i_t wc[SC_NU];
#pragma unroll
    for (int i = 0; i < SC_NU; ++i) {
        const int u = i < nu ? u0 + i : u0;                      // (the fifth slot of a four-unit wave is never used)
        const unsigned char *p = w2 + ((size_t)(cth * 18 + u) * 64 + lane) * 64;
        wa[i][0] = *reinterpret_cast<const h8_t *>(p);
        wa[i][1] = *reinterpret_cast<const h8_t *>(p + 16);
        wc[i] = sfd2_cat8(*reinterpret_cast<const h8_t *>(p + 32), *reinterpret_cast<const h8_t *>(p + 48));
    }
    // ---- conv1a filter fragments (this wave's 32-channel half), hi and lo: re-requested per tile right after phase 2 (from
    // L2, 96 B per lane) instead of held across it -- phase 2 needs the 24 registers for its accumulators
    h8_t a1h[3], a1l[3];
#define SC_LOAD_A1()                                                                                       \
    {                                                                                                      \
        int ao_ = (cth * 3 * 64 + lane) * 8;                                                               \
        asm volatile("" : "+v"(ao_));   /* (opaque: not hoisted back out of the tile loop) */               \
        _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                                 \
            a1h[ky] = *reinterpret_cast<const h8_t *>(w1 + ao_ + ky * 512);                                \
            a1l[ky] = *reinterpret_cast<const h8_t *>(w1 + ao_ + 3072 + ky * 512);                         \
        }                                                                                                  \
    }
    SC_LOAD_A1()

    // ---- image patch of a tile: raw values fetched one tile ahead, normalised and split when they are written to LDS
    unsigned int pr[SC_IPT][3];
    unsigned pr_inside = 0;
    int fpy[SC_IPT], fpx[SC_IPT];
#pragma unroll
    for (int k = 0; k < SC_IPT; ++k) {
        const int p = tid + k * SC_NT;
        fpy[k] = p / SC_IW;
        fpx[k] = p - fpy[k] * SC_IW;
    }
#define SC_FETCH_IMG(tile_)                                                                                \
    {                                                                                                      \
        const int ftx = (tile_) % tiles_x, fty = (tile_) / tiles_x;                                        \
        const int fy0 = 2 * (fty * SC_TH) - 2, fx0 = 2 * (ftx * SC_TW) - 2;                                \
        pr_inside = 0;                                                                                     \
        _Pragma("unroll") for (int k = 0; k < SC_IPT; ++k) {                                               \
            const int p = tid + k * SC_NT;                                                                 \
            const int iy = fy0 + fpy[k], ix = fx0 + fpx[k];                                                \
            unsigned int r = 0u, g = 0u, b = 0u;                                                           \
            if (p < SC_IH * SC_IW && iy >= 0 && iy < H && ix >= 0 && ix < W) {                             \
                const size_t o = (size_t)iy * W + ix;                                                      \
                pr_inside |= 1u << k;                                                                      \
                if (normalise & 2) { /* uint8 HWC ingest (extract_localization.py:165-186) */              \
                    const unsigned char *u = reinterpret_cast<const unsigned char *>(img) + o * 3;         \
                    const int sw = (normalise & 4) ? 2 : 0;                                                \
                    r = u[sw]; g = u[1]; b = u[2 - sw];                                                    \
                } else {                                                                                   \
                    const unsigned int *iu = reinterpret_cast<const unsigned int *>(img);                  \
                    r = iu[o]; g = iu[plane + o]; b = iu[2 * plane + o];                                   \
                }                                                                                          \
            }                                                                                              \
            pr[k][0] = r; pr[k][1] = g; pr[k][2] = b;                                                      \
        }                                                                                                  \
    }
#define SC_STORE_IMG()                                                                                     \
    _Pragma("unroll") for (int k = 0; k < SC_IPT; ++k) {                                                   \
        const int p = tid + k * SC_NT;                                                                     \
        float r, g, b;                                                                                     \
        if (normalise & 2) { r = (float)pr[k][0]; g = (float)pr[k][1]; b = (float)pr[k][2]; }              \
        else { r = __uint_as_float(pr[k][0]); g = __uint_as_float(pr[k][1]); b = __uint_as_float(pr[k][2]); } \
        if (pr_inside & (1u << k)) {                                                                       \
            if (normalise & 2) {                                                                           \
                r = sc_div_const(r, 255.0f, 1.0f / 255.0f); g = sc_div_const(g, 255.0f, 1.0f / 255.0f);    \
                b = sc_div_const(b, 255.0f, 1.0f / 255.0f);                                                \
            }                                                                                              \
            if (normalise & 1) {                                                                           \
                r = sc_div_const(__fsub_rn(r, 0.485f), 0.229f, 1.0f / 0.229f);                             \
                g = sc_div_const(__fsub_rn(g, 0.456f), 0.224f, 1.0f / 0.224f);                             \
                b = sc_div_const(__fsub_rn(b, 0.406f), 0.225f, 1.0f / 0.225f);                             \
            }                                                                                              \
        }                                                                                                  \
        if (p < SC_IH * SC_IW) {                                                                           \
            h4_t hh, hl;                                                                                   \
            hh[0] = (half_t)r; hh[1] = (half_t)g; hh[2] = (half_t)b; hh[3] = (half_t)0.0f;                 \
            hl[0] = (half_t)(r - (float)hh[0]); hl[1] = (half_t)(g - (float)hh[1]); hl[2] = (half_t)(b - (float)hh[2]); hl[3] = (half_t)0.0f; \
            *reinterpret_cast<h4_t *>(IMh + p * 4) = hh;                                                   \
            *reinterpret_cast<h4_t *>(IMl + p * 4) = hl;                                                   \
        }                                                                                                  \
    }

    // ---- phase 1 geometry (the same for every tile): unit = (32-pixel block of the region, this wave's channel half)
    constexpr int P1_UNITS = ((SC_RP + 31) / 32 + 3) / 4;
    const int p1_n = ((SC_RP + 31) / 32 - kg + 3) / 4;
    int p1_im[P1_UNITS], p1_x[P1_UNITS], p1_yx[P1_UNITS];
#pragma unroll
    for (int i = 0; i < P1_UNITS; ++i) {
        const int p = (kg + 4 * i) * 32 + lrow;
        // the lanes past the region's last pixel (the last block has 9 of 32) compute that last pixel again and store the
        // same bytes to the same place: no predicate around the stores, so the four channel quads of a unit are ONE
        // straight-line block (predicated, each quad was its own exec-masked block with its LDS round trips exposed)
        const int pc = p < SC_RP ? p : SC_RP - 1;
        const int ry = pc / SC_RW, rx = pc - ry * SC_RW;
        p1_yx[i] = (ry << 8) | rx;
        p1_im[i] = (ry * SC_IW + rx + 2 * lhi) * 4;                      // halfs
        // byte offset of this lane's 8 bytes of channel quad 0 in the pixel's record, swizzle of the record folded in per quad below
        p1_x[i] = ((pc ^ ((pc >> 4) & 1)) * 128 + 8 * lhi) | (SC_SWZ(pc) << 20);
    }

    int tile = blockIdx.x;
    SC_FETCH_IMG(tile)
    SC_STORE_IMG()
    if (tile + (int)gridDim.x < n_tiles) SC_FETCH_IMG(tile + (int)gridDim.x)
    SFD2_BARRIER_DRAIN();
    // (the first tile's conv1a filters have landed behind that drain; tell the wait-count bookkeeping so -- merged with the
    // loop's back edge, a still-pending prologue load would turn the first MFMA of EVERY tile into a wait that also covers
    // the previous tile's stores and the image request in flight)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) asm volatile("" : "+v"(a1h[ky]), "+v"(a1l[ky]));

    int tcount = 0;
    (void)tcount;
    for (;; ++tcount) {
        SC_STAMP(0)
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int oy0 = ty * SC_TH, ox0 = tx * SC_TW;
        const int ry0 = 2 * oy0 - 1, rx0 = 2 * ox0 - 1;

        float mx1 = 0.0f, mx2 = 0.0f;
        // ---- phase 1: conv1a -> X1h / X1c
        // -DSFD2_STEMC_P1MODE (round 5 experiment): the two waves of a SIMD (w and w + 4) run phase 1's units in lockstep -- both in their MFMA chain, then both
        // in their epilogue -- so a SIMD's time is MFMA time + VALU time.  1: priority 1 for a unit's MFMA chain; 2: waves 0..3 at priority 2 for the whole
        // phase; 3: waves 4..7 start the phase ~320 cycles late
#if SFD2_STEMC_P1MODE == 2
        if (wave < 4) __builtin_amdgcn_s_setprio(2);
#elif SFD2_STEMC_P1MODE == 3
        if (wave >= 4) __builtin_amdgcn_s_sleep(5);
#endif
        // conv1a's scale / shift of this wave's channels: read once per tile (they must not be live across phase 2); PIPE: once per unit -- the
        // second accumulator and the next unit's fragments need the 32 registers
#define SC_LOAD_S1()                                                                                       \
        {                                                                                                  \
            int sso = cth * 32 + 4 * lhi;                                                                  \
            asm volatile("" : "+v"(sso));                                                                  \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                \
                s1[q] = sfd2_lds_f4(SS + sso + 8 * q);                                                     \
                h1[q] = sfd2_lds_f4(SS + 64 + sso + 8 * q);                                                \
            }                                                                                              \
        }
        // -DSFD2_STEMC_P1PIPE (round 5, the fp6 instantiation): the MFMA chain of unit i + 1 is issued BETWEEN the pieces of unit i's epilogue (a second
        // accumulator), so that a wave's own MFMAs run under its own ~115 epilogue instructions -- with two waves per SIMD a wave issues one vector
        // instruction per ~7.5 cycles, a lone wave cannot fill the SIMD while its partner sits in an MFMA chain, and in practice both sit in their chains
        // and then in their epilogues together (tools/probe/valu_cost.hip, profiles/r05j_*).  The last unit's chain is issued by every wave (the four-unit
        // waves compute the region's last pixel block again and drop it).
#ifdef SFD2_STEMC_P1PIPE
        constexpr bool PIPE = O6 && !X3;
#else
        constexpr bool PIPE = false;
#endif
        float4 s1[4], h1[4];
        if constexpr (!PIPE) SC_LOAD_S1()
        // the image fragments of one kernel row of a unit (imo_: the unit's opaque offset register)
#define SC_P1_FRAG(imo_, ky_, bh_, bl_)                                                                    \
        {                                                                                                  \
            const int o = (imo_) + (ky_) * (SC_IW * 4);                                                    \
            {                                                                                              \
                const h4_t lo = *reinterpret_cast<const h4_t *>(IMh + o);                                  \
                const h4_t hi = *reinterpret_cast<const h4_t *>(IMh + o + 4);                              \
                bh_[0] = lo[0]; bh_[1] = lo[1]; bh_[2] = lo[2]; bh_[3] = lo[3];                            \
                bh_[4] = hi[0]; bh_[5] = hi[1]; bh_[6] = hi[2]; bh_[7] = hi[3];                            \
            }                                                                                              \
            {                                                                                              \
                const h4_t lo = *reinterpret_cast<const h4_t *>(IMl + o);                                  \
                const h4_t hi = *reinterpret_cast<const h4_t *>(IMl + o + 4);                              \
                bl_[0] = lo[0]; bl_[1] = lo[1]; bl_[2] = lo[2]; bl_[3] = lo[3];                            \
                bl_[4] = hi[0]; bl_[5] = hi[1]; bl_[6] = hi[2]; bl_[7] = hi[3];                            \
            }                                                                                              \
        }
        // one unit's MFMA chain in one run
#define SC_P1_CHAIN(i_, acc_)                                                                              \
        {                                                                                                  \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc_[r] = 0.0f;                                 \
            /* per-lane offsets are recomputed from one opaque register per unit: hoisted out of the tile loop (they are all */ \
            /* tile-invariant) they are ~100 registers, spilled and reloaded behind s_waitcnt vmcnt(0) */  \
            int imo = p1_im[i_];                                                                           \
            asm volatile("" : "+v"(imo));                                                                  \
            SC_P1_PRIO(1)                                                                                  \
            _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                             \
                h8_t bh, bl;                                                                               \
                SC_P1_FRAG(imo, ky, bh, bl)                                                                \
                acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l[ky], bh, acc_, 0, 0, 0);                 \
                acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[ky], bl, acc_, 0, 0, 0);                 \
                acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[ky], bh, acc_, 0, 0, 0);                 \
            }                                                                                              \
            SC_P1_PRIO(0)                                                                                  \
        }
#if SFD2_STEMC_P1MODE == 1
#define SC_P1_PRIO(p_) __builtin_amdgcn_s_setprio(p_);
#else
#define SC_P1_PRIO(p_)
#endif
        f32x16_t accs[2];
        h8_t fbh[2], fbl[2];                      // PIPE: the next unit's fragments, two kernel rows in flight
        if constexpr (PIPE) SC_P1_CHAIN(0, accs[0])
#pragma unroll
        for (int i = 0; i < P1_UNITS; ++i) {
            if (i + 1 < P1_UNITS || i < p1_n) {      // (every wave has P1_UNITS - 1 or P1_UNITS units: only the last one is behind a branch)
                f32x16_t &acc = accs[PIPE ? (i & 1) : 0];
                if constexpr (!PIPE) SC_P1_CHAIN(i, acc)
                // conv1b zero-pads conv1a's OUTPUT: region pixels outside the image are zeros, not conv1a(0)
                const int gy = ry0 + (p1_yx[i] >> 8), gx = rx0 + (p1_yx[i] & 255);
                const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
                // (interior tiles: a wave-uniform flag lets the selects below fold away)
                const bool all_inside = __builtin_amdgcn_ballot_w64(!inside) == 0;
                int xo = p1_x[i] & 0xFFFFF, xsw = (p1_x[i] >> 20) << 4;
                asm volatile("" : "+v"(xo), "+v"(xsw));
                if constexpr (O6 && !X3) {
                    uint2 hv4[4];
                    uint4 r0, r1;
                    if constexpr (PIPE) {
                        const bool more = i + 1 < P1_UNITS;       // (a constant once the unit loop is unrolled)
                        f32x16_t &nacc = accs[(i + 1) & 1];
                        int imo = p1_im[more ? i + 1 : i];
                        asm volatile("" : "+v"(imo));
                        if (more) SC_P1_FRAG(imo, 0, fbh[0], fbl[0])
                        SC_LOAD_S1()
                        sc_epi16_fp6_piped(acc, s1, h1, hv4, r0, r1, mx1, all_inside || inside, [&](const int k) {
                            if (more) {
                                const int ky = k / 3, j = k - 3 * ky;
                                if (j == 0 && ky < 2) SC_P1_FRAG(imo, ky + 1, fbh[(ky + 1) & 1], fbl[(ky + 1) & 1])
                                f32x16_t c = nacc;
                                if (k == 0) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
                                }
                                nacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(j == 0 ? a1l[ky] : a1h[ky], j == 1 ? fbl[ky & 1] : fbh[ky & 1], c, 0, 0, 0);
                            }
                        });
                    } else
                        sfd2_epi16_fp6(acc, s1, h1, 0.0f, hv4, r0, r1, mx1, all_inside || inside);
                    if (!all_inside && !inside) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) hv4[q] = make_uint2(0u, 0u);
                        r0 = make_uint4(0u, 0u, 0u, 0u); r1 = r0;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2 *>(X1h + xo + (((cth * 4 + q) << 4) ^ xsw)) = hv4[q];
                    const int xb = xo & ~8;             // the record's base: half-record lhi = the chunk's 16-byte slots lhi and 2 + lhi
                    *reinterpret_cast<uint4 *>(X1c + xb + (((cth * 4 + lhi) << 4) ^ xsw)) = r0;
                    *reinterpret_cast<uint4 *>(X1c + xb + (((cth * 4 + 2 + lhi) << 4) ^ xsw)) = r1;
                } else
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 hv, cv;
                    if (X3) sc_x3_epi4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], s1[q], h1[q], hv, cv);
                    else sfd2_epi4<false>(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], s1[q], h1[q], s1[q], 0.0f, hv, cv, mx1, all_inside || inside);
                    if (!all_inside && !inside) { hv = make_uint2(0u, 0u); cv = make_uint2(0u, 0u); }
                    {
                        const int o = xo + (((cth * 4 + q) << 4) ^ xsw);
                        *reinterpret_cast<uint2 *>(X1h + o) = hv;
                        *reinterpret_cast<uint2 *>(X1c + o) = cv;
                    }
                }
            }
#ifdef SFD2_STEMC_SCHEDB
            __builtin_amdgcn_sched_barrier(0);   // (experiment: units not interleaved)
#endif
        }
#if SFD2_STEMC_P1MODE == 2
        __builtin_amdgcn_s_setprio(0);
#endif
        if (!X3) { const unsigned int wb = sfd2_wave_max_bits(mx1); smax1 = wb > smax1 ? wb : smax1; }
        const int next = tile + (int)gridDim.x, next2 = next + (int)gridDim.x;
        const bool has_next = next < n_tiles;
        SC_STAMP(1)
        SC_LDS_BARRIER();                     // X1 complete; IM is free from here on
        SC_STAMP(2)

        // ---- phase 2: partial sums of conv1b over this wave's units, all four output rows
        f32x16_t acc2[SC_TH];
#pragma unroll
        for (int r = 0; r < SC_TH; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[r][e] = 0.0f;
        int l2 = 2 * lrow;
        asm volatile("" : "+v"(l2));          // (the 40 fragment addresses of phase 2 are tile-invariant too)
#pragma unroll
        for (int i = 0; i < SC_NU; ++i) {
            if (i < nu) {                     // wave-uniform
                const int u = u0 + i, tap = u >> 1, ih = u & 1;
                const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                for (int r = 0; r < SC_TH; ++r) {
                    const int q = (2 * r + ky) * SC_RW + l2 + kx;
                    const int ro = (q ^ ((q >> 4) & 1)) * 128;
                    // (the pixel's parity is the tap's: a constant, folded into the slot constants -- as `SC_SWZ(q)` it was two more vector instructions per read)
                    const int bsw = (q >> 1) & 7, par = SC_SWZ_PAR((2 * r + ky) * SC_RW + kx);
                    const int o0 = ro + (((((ih * 4) ^ par) + lhi) ^ bsw) << 4), o1 = ro + (((((ih * 4 + 2) ^ par) + lhi) ^ bsw) << 4);
                    const h8_t b0 = *reinterpret_cast<const h8_t *>(X1h + o0);
                    const h8_t b1 = *reinterpret_cast<const h8_t *>(X1h + o1);
                    acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][0], b0, acc2[r], 0, 0, 0);
                    acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][1], b1, acc2[r], 0, 0, 0);
                    if (!X3) {
                        const v8i_t bc = sfd2_cat8(*reinterpret_cast<const h8_t *>(X1c + o0), *reinterpret_cast<const h8_t *>(X1c + o1));
                        if constexpr (O6) acc2[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wc[i], bc, acc2[r], 2, SFD2_PIX6_BLGP, 0, wc[i][6], 0, bc[6]);
                        else acc2[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wc[i], bc, acc2[r], 0, 0, 0, sa, 0, 0x7f7f7f7f);
                    }
                }
            }
        }
        if (X3) {
            // the cross terms, at 2^11 times their weight, into the same accumulators
#pragma unroll
            for (int r = 0; r < SC_TH; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[r][e] *= 2048.0f;
            int l2b = 2 * lrow;
            asm volatile("" : "+v"(l2b));
#pragma unroll
            for (int i = 0; i < SC_NU; ++i) {
                if (i < nu) {
                    const int u = u0 + i, tap = u >> 1, ih = u & 1;
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int r = 0; r < SC_TH; ++r) {
                        const int q = (2 * r + ky) * SC_RW + l2b + kx;
                        const int ro = (q ^ ((q >> 4) & 1)) * 128;
                        const int bsw = (q >> 1) & 7, par = SC_SWZ_PAR((2 * r + ky) * SC_RW + kx);
                        const int o0 = ro + (((((ih * 4) ^ par) + lhi) ^ bsw) << 4), o1 = ro + (((((ih * 4 + 2) ^ par) + lhi) ^ bsw) << 4);
                        const h8_t b0 = *reinterpret_cast<const h8_t *>(X1h + o0);
                        const h8_t b1 = *reinterpret_cast<const h8_t *>(X1h + o1);
                        const h8_t l0 = *reinterpret_cast<const h8_t *>(X1c + o0);
                        const h8_t l1 = *reinterpret_cast<const h8_t *>(X1c + o1);
                        acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(wc[i], 0), b0, acc2[r], 0, 0, 0);
                        acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(wc[i], 1), b1, acc2[r], 0, 0, 0);
                        acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][0], l0, acc2[r], 0, 0, 0);
                        acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[i][1], l1, acc2[r], 0, 0, 0);
                    }
                }
            }
        }
        // (the scaled MFMA is a pure node to instruction selection: pin the partial sums in front of the barrier)
#pragma unroll
        for (int r = 0; r < SC_TH; ++r) asm volatile("" : "+v"(acc2[r]));
        if (has_next) SC_LOAD_A1()            // conv1a filters of the next tile's phase 1
        SC_STAMP(3)
        SC_LDS_BARRIER();                     // every wave is done reading X1
        SC_STAMP(4)

        // ---- phase 3: this wave's partials of all four rows -> LDS (its own row too: phase 4 then reads four tiles in a
        // fixed order with static register indices)
#pragma unroll
        for (int r = 0; r < SC_TH; ++r) {
            float *pt = PT + (size_t)((cth * 4 + r) * 4 + kg) * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(pt + q * 256) = make_float4(acc2[r][4 * q], acc2[r][4 * q + 1], acc2[r][4 * q + 2], acc2[r][4 * q + 3]);
        }
        SC_STAMP(5)
        SC_LDS_BARRIER();                     // partials visible
        SC_STAMP(6)

        // The conv1a filters requested after phase 2 are taken into registers HERE, where they are the only vector-memory
        // operations in flight: left to the first MFMA of the next tile's phase 1, hipcc waits for them with vmcnt(0) behind
        // this tile's output stores and the image request of the tile after next (one exposed memory latency per tile).
        if (has_next) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) asm volatile("" : "+v"(a1h[ky]), "+v"(a1l[ky]));
        }
        // ---- phase 4: row kg of this wave's channel half = partials of groups 0, 1, 2, 3 added in that order
        f32x16_t tot;
        {
            const float *pt = PT + (size_t)((cth * 4 + kg) * 4) * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v0 = sfd2_lds_f4(pt + q * 256), v1 = sfd2_lds_f4(pt + 1024 + q * 256);
                const float4 v2 = sfd2_lds_f4(pt + 2048 + q * 256), v3 = sfd2_lds_f4(pt + 3072 + q * 256);
                tot[4 * q + 0] = ((v0.x + v1.x) + v2.x) + v3.x;
                tot[4 * q + 1] = ((v0.y + v1.y) + v2.y) + v3.y;
                tot[4 * q + 2] = ((v0.z + v1.z) + v2.z) + v3.z;
                tot[4 * q + 3] = ((v0.w + v1.w) + v2.w) + v3.w;
            }
        }

        SC_STAMP(7)
        if (has_next) SC_STORE_IMG()          // the next tile's patch (requested a tile ago) -> IM, in front of the output stores
#ifndef SFD2_STEMC_FETCH_LATE
        // ... and the patch of the tile after next is requested HERE, in front of this tile's output stores: issued behind them (as until round 4)
        // the request sat 1.3-4.3k cycles in the vector-memory queue (profiles/r04e_stemc_trace.txt, column "fetch issue")
        if (has_next && next2 < n_tiles) SC_FETCH_IMG(next2)
#endif
        SC_STAMP(8)
        const int oy = oy0 + kg, ox = ox0 + lrow;
        const bool inb = oy < H2 && ox < W2;
        const size_t ob = ((size_t)(inb ? oy : 0) * W2 + (inb ? ox : 0)) * 64;
        if constexpr (O6 && !X3) {
            float4 s4[4], h4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s4[q] = sfd2_lds_f4(SS + 128 + cth * 32 + 8 * q + 4 * lhi);
                h4[q] = sfd2_lds_f4(SS + 192 + cth * 32 + 8 * q + 4 * lhi);
            }
            uint2 hv4[4];
            uint4 r0, r1;
            sfd2_epi16_fp6(tot, s4, h4, 0.0f, hv4, r0, r1, mx2, inb);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const auto t0 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].x, hv4[2 * m + 1].x, false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].y, hv4[2 * m + 1].y, false, false);
                if (inb) *reinterpret_cast<uint4 *>(out + ob + cth * 32 + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            }
            if (inb) {
                *reinterpret_cast<uint4 *>(out_c + ob + cth * 32 + 8 * lhi) = r0;
                *reinterpret_cast<uint4 *>(out_c + ob + cth * 32 + 16 + 8 * lhi) = r1;
            }
        } else
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            uint2 pk[2], ck[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 2 * m + j;
                const int c0 = cth * 32 + 8 * q + 4 * lhi;
                const float4 s = sfd2_lds_f4(SS + 128 + c0);
                const float4 h = sfd2_lds_f4(SS + 192 + c0);
                if (X3) sc_x3_epi4(tot[4 * q + 0] * (1.0f / 2048.0f), tot[4 * q + 1] * (1.0f / 2048.0f), tot[4 * q + 2] * (1.0f / 2048.0f),
                                   tot[4 * q + 3] * (1.0f / 2048.0f), s, h, pk[j], ck[j]);
                else sfd2_epi4<false>(tot[4 * q + 0], tot[4 * q + 1], tot[4 * q + 2], tot[4 * q + 3], s, h, s, 0.0f, pk[j], ck[j], mx2, inb);
            }
            const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
            const auto u0s = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
            const auto u1s = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
            if (inb) {
                const size_t o = ob + cth * 32 + 8 * (2 * m + lhi);
                *reinterpret_cast<uint4 *>(out + o) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                *reinterpret_cast<uint4 *>(out_c + o) = make_uint4(u0s[0], u1s[0], u0s[1], u1s[1]);
            }
        }
        if (!X3) { const unsigned int wb = sfd2_wave_max_bits(mx2); smax2 = wb > smax2 ? wb : smax2; }
        SC_STAMP(9)
        if (!has_next) break;
#ifdef SFD2_STEMC_FETCH_LATE
        if (next2 < n_tiles) SC_FETCH_IMG(next2)
#endif
        SC_STAMP(10)
        SC_LDS_BARRIER();                     // IM complete; every wave is done with the partials (phase 1 writes X1 again)
        SC_STAMP(11)
        tile = next;
    }
    if (!X3 && range != nullptr) {
        sfd2_range_commit(range + SFD2_RS_CONV1A * SFD2_RANGE_SUB, smax1);
        sfd2_range_commit(range + SFD2_RS_CONV1B * SFD2_RANGE_SUB, smax2);
    }
#undef SC_LOAD_A1
#undef SC_P1_CHAIN
#undef SC_P1_FRAG
#undef SC_LOAD_S1
#undef SC_P1_PRIO
#undef SC_FETCH_IMG
#undef SC_STORE_IMG
}

// sbyte < 0: the X3 instantiation (SFD2_PREC_F16X3): w2's second halves are the filters' lo' fragments, out / out_c the hi / lo' planes
void launch_fused_stem_c(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *w1, const float *sc1,
                         const float *sh1, const void *w2, const float *sc2, const float *sh2, half_t *out, half_t *out_c,
                         int H2, int W2, int sbyte, unsigned int *range, int fmt6 /* bit 1: fp6 records inside and out; w2 = the fp6 fragment array */)
{
    static bool attr_done = false;
    static int slots = 256;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stem_c_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stem_c_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fused_stem_c_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;   // 159 KB of LDS: one resident block per CU
        attr_done = true;
    }
    const int tiles_x = (W2 + SC_TW - 1) / SC_TW, tiles_y = (H2 + SC_TH - 1) / SC_TH;
    const int n_tiles = tiles_x * tiles_y;
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    if (sbyte < 0)
        hipLaunchKernelGGL(fused_stem_c_kernel<true>, dim3(grid), dim3(SC_NT), SC_LDS, st, img, H, W, normalise, w1, sc1, sh1,
                           reinterpret_cast<const unsigned char *>(w2), sc2, sh2, out, out_c, H2, W2, tiles_x, n_tiles, 0, nullptr);
    else if (fmt6 & 2)
        hipLaunchKernelGGL((fused_stem_c_kernel<false, true>), dim3(grid), dim3(SC_NT), SC_LDS, st, img, H, W, normalise, w1, sc1, sh1,
                           reinterpret_cast<const unsigned char *>(w2), sc2, sh2, out, out_c, H2, W2, tiles_x, n_tiles,
                           (sbyte & 255) * 0x01010101, range);
    else
        hipLaunchKernelGGL(fused_stem_c_kernel<false>, dim3(grid), dim3(SC_NT), SC_LDS, st, img, H, W, normalise, w1, sc1, sh1,
                           reinterpret_cast<const unsigned char *>(w2), sc2, sh2, out, out_c, H2, W2, tiles_x, n_tiles,
                           (sbyte & 255) * 0x01010101, range);
#ifdef SFD2_STEMC_TRACE
    {
        static int dumps = 0;
        if (H >= 1000 && ++dumps == 40) {
            (void)hipStreamSynchronize(st);
            static unsigned long long h[8][16][12];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stemc_trace), sizeof(h));
            fprintf(stderr, "stemc columns: phase 1 | B1 wait | phase 2 (+A1 reload issue) | B2 wait | phase 3 partial writes | B3 wait | phase 4 sum | image -> LDS | epilogue + stores | fetch issue | B4 wait\n");
            for (int w = 0; w < 8; ++w)
                for (int t = 4; t < 8; ++t) {
                    fprintf(stderr, "stemc wave %d tile %2d (phase 1 starts %5lld after wave 0's):", w, t, (long long)(h[w][t][0] - h[0][t][0]));
                    for (int k = 1; k < 12; ++k) fprintf(stderr, " %6lld", (long long)(h[w][t][k] - h[w][t][k - 1]));
                    fprintf(stderr, "  (tile %lld)\n", (long long)(h[w][t + 1][0] - h[w][t][0]));
                }
        }
    }
#endif
}
