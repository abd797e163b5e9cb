// Fused ResBlock (nets/sfd2.py:25-55): conv1 1x1 + bn1 + ReLU -> grouped 3x3 (groups = 32) + bn2 + ReLU ->
// conv3 1x1 + bn3 + residual + ReLU in ONE kernel.  Unfused, the block moves seven 61 MB tensor passes through HBM
// (x, t1, t1, t2, t2, x, out at 1600x1200) and is bound by exactly that; here t1 and t2 never leave the CU:
// HBM sees x once (+7 % column halo, + two halo rows per segment) and out once.
//
// Work item = a 30-pixel-wide column strip x a segment of rows, marched top to bottom one image row at a time:
//   x row r   (32 px incl. the 1-px column halo, 16 KB)  -> LDS ring by direct-to-LDS copies, one row ahead
//   conv1(r)  : t1 row r = relu(bn1(W1 . x))             -> LDS ring of three t1 rows (zero outside the image:
//                                                            the grouped conv zero-pads t1, not x)
//   gconv(r-1): t2 row from t1 rows r-2, r-1, r           -> LDS (one row)
//   conv3(r-1): out = relu(bn3(W3 . t2) + x row r-1)      -> HBM; the residual is read from the x ring in LDS
// Block = 8 waves: wave w owns channels [32w, 32w + 32) at EVERY stage, so its W1 and W3 slices (2 x 64 VGPRs) and
// its grouped-conv fragments (40 VGPRs) stay in registers for the whole kernel, and the t1 ring needs no block
// barrier at all (a wave's grouped conv only reads channels that wave wrote).  (A 4-wave / 512-register variant
// with 64 channels per wave spilled its 336 weight registers and ran 135-165 us per block.)  Two barriers
// per row: x row landed (counted vmcnt: the next row's copies stay in flight) and t2 row complete.
// LDS layouts (512 B per pixel): the t1 ring XOR-swizzles the 16-byte slot index with (pixel & 31); the x ring and
// the t2 row, which feed the 1x1 convolutions' B fragments, pad every PAIR of pixels to 1056 B and XOR only slot bit 0
// with the pixel's parity: equally conflict-free for the fragment reads, but a lane's 16 k-slice addresses are then
// base + 32 * kk -- immediates, no address arithmetic (the kernel was VALU-bound: 610 VALU instructions per row).
#include "sfd2_internal.h"
#include <stdlib.h>
#include <stdio.h>

#define RB_NT 512
#define RB_SW 30                    // output columns per strip (32 with the halo = one MFMA pixel tile)
#ifndef RB_PF
#define RB_PF 2                     // k-slice fragments in flight ahead of the 1x1 convolutions' MFMAs
#endif
#define RB_NX 4                     // x rows in the ring: r - 1 (residual), r, the one in flight, and one of slack that replaces a barrier
#define RB_GW_BYTES (256 * 9 * 8 * 2) // grouped-conv filters, compact [oc][tap][8 in] fp16 (36 KB)
#define RB_ROW (32 * 512)           // bytes of one 32-pixel row of 256 fp16 channels (t1 ring: XOR-swizzled records)
#define RB_PROW (16 * 1056)         // x ring / t2 row: pixel PAIRS of 1024 B + 32 B pad, slot bit 0 XORed with the pixel's parity

// -DSFD2_RB_TRACE: cycle stamps of one block's waves 0 and 4 at the section boundaries of every row, printed by the
// launcher after the 40th launch (how the per-row budget in DESIGN.md was measured)
#ifdef SFD2_RB_TRACE
__device__ unsigned long long g_rb_trace[2][64][8];   // [wave 0 | 4][row][stamp]
#define RB_STAMP(k_)                                                                          \
    if (blockIdx.x == gridDim.x / 2 && (wave == 0 || wave == 4) && lane == 0 && r - (ya - 1) < 64) \
        g_rb_trace[wave >> 2][r - (ya - 1)][k_] = __builtin_readcyclecounter();
#else
#define RB_STAMP(k_)
#endif
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ h4_t rb_cvt4(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// scale / shift of 4 consecutive channels starting at base + 4 * lhi, base wave-uniform: scalar loads of both halves,
// selected per lane half.  NOT used: measured slower than the LDS reads in both 1x1 epilogues (conv4.x 73 -> 82 us: the
// scalar-load latency plus eight v_cndmask per float4 pair cost more than the LDS round trips they replace).
// 16-byte LDS read of four floats, TYPED LIKE THE MFMA FRAGMENT READS.  hipcc orders LDS reads against in-flight
// direct-to-LDS copies by type-based alias analysis: a float-typed ds_read "may alias" the copies' destination, so the
// compiler drains s_waitcnt vmcnt(0) in front of it -- i.e. in front of every epilogue's scale / shift reads, once the next
// row's copies (and the previous row's output stores) are in flight: three full memory-latency stalls per row.  Read as
// _Float16 x 8 like the fragments, the same bytes carry no such wait (conv4.x 77 -> 70 us at 1600x1200).
__device__ __forceinline__ float4 rb_lds4(const float *p)
{
    const h8_t raw = *reinterpret_cast<const h8_t *>(reinterpret_cast<const unsigned char *>(p));
    float4 r;
    __builtin_memcpy(&r, &raw, 16);
    return r;
}

// the same through an LDS byte offset: base register + immediate (one address VGPR for all of a wave's scale / shift reads)
__device__ __forceinline__ float4 rb_lds4o(const unsigned char *lds, unsigned off)
{
    const h8_t raw = *reinterpret_cast<const h8_t *>(lds + off);
    float4 r;
    __builtin_memcpy(&r, &raw, 16);
    return r;
}

__device__ __forceinline__ float4 rb_ss(const float *__restrict__ p, int base, int lhi)
{
    const float4 lo = *reinterpret_cast<const float4 *>(p + base), hi = *reinterpret_cast<const float4 *>(p + base + 4);
    return lhi ? hi : lo;
}

__global__ __launch_bounds__(RB_NT, 2)
void resblock_kernel(const half_t *__restrict__ x, int H, int W,
                     const half_t *__restrict__ w1 /*[256][256]*/, const float *__restrict__ sc1, const float *__restrict__ sh1,
                     const half_t *__restrict__ wg /*[256 oc][9 taps][8 in] grouped-conv filters*/,
                     const float *__restrict__ sc2, const float *__restrict__ sh2,
                     const half_t *__restrict__ w3 /*[256][256]*/, const float *__restrict__ sc3, const float *__restrict__ sh3,
                     half_t *__restrict__ out, int strips, int rows_per_item, const half_t *__restrict__ zero_page)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *XR = smem;                               // [RB_NX] pair-padded rows
    unsigned char *T2 = XR + RB_NX * RB_PROW;               // [2] pair-padded rows
    unsigned char *T1 = T2 + 2 * RB_PROW;                   // [3][32][512]
    float *SS = reinterpret_cast<float *>(T1 + 3 * RB_ROW);   // sc1 sh1 sc2 sh2 sc3 sh3, 256 each

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int strip = blockIdx.x % strips, seg = blockIdx.x / strips;
    const int ya = seg * rows_per_item;
    int yb = ya + rows_per_item;            // output rows [ya, yb)
    if (yb > H) yb = H;
    if (ya >= yb) return;
    const int col0 = strip * RB_SW - 1;     // image column of strip pixel 0

    // ---- filters of this wave (channels 32 * wave .. + 31), resident for the whole kernel
    h8_t a1[16], a3[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const size_t o = (size_t)(wave * 32 + lrow) * 256 + kk * 16 + lhi * 8;
        a1[kk] = *reinterpret_cast<const h8_t *>(w1 + o);
        a3[kk] = *reinterpret_cast<const h8_t *>(w3 + o);
    }
    for (int t = tid; t < 256; t += RB_NT) {
        SS[t] = sc1[t]; SS[256 + t] = sh1[t]; SS[512 + t] = sc2[t]; SS[768 + t] = sh2[t]; SS[1024 + t] = sc3[t]; SS[1280 + t] = sh3[t];
    }

    // x row r -> ring slot: 16 one-KB chunks (2 pixels each), 2 per wave.  Rows / columns outside the image read
    // the zero page (their t1 is masked to zero anyway; the copies are issued regardless so that every wave
    // always has the same number of vector-memory operations in flight -- the waits below count them).
#define RB_ISSUE_X(r_)                                                                                     \
    {                                                                                                      \
        unsigned char *dst = XR + (((r_) + RB_NX) % RB_NX) * RB_PROW;                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
            const int ch = wave * 2 + i;                           /* pixel pair */                        \
            const int col = col0 + ch * 2 + lhi;                                                           \
            const bool ok = (r_) >= 0 && (r_) < H && col >= 0 && col < W;                                  \
            const half_t *src = ok ? x + ((size_t)(r_)*W + col) * 256 + ((lrow ^ lhi) << 3) : zero_page + (lrow << 3); \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)(dst + ch * 1056), 16, 0, 0); \
        }                                                                                                  \
    }
#define RB_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    RB_ISSUE_X(ya - 1)
    SFD2_BARRIER_DRAIN();    // GW and the first row complete

    const int n = lrow;                       // this lane's strip pixel in the 32x32 MFMAs
    const int pp_base = (n >> 1) * 1056 + (n & 1) * 512;               // its record in a pair-padded row
    const int bfrag = pp_base + ((lhi ^ (n & 1)) << 4);                // + 32 * kk = its B fragment of k slice kk
    const int ncol = col0 + n;
    // Loop-invariant per-lane offsets, kept to a few registers by hand.  Left to itself hipcc hoists every (tap, pixel
    // tile, channel pair) address of the grouped conv and every scale / shift / t1-write address out of the row loop --
    // ~45 registers on top of the 128 of W1 / W3 -- and then has none left to keep more than one LDS read in flight: the
    // grouped conv ran as 30 dependent LDS round trips per row (2200 of the row's 7700 cycles), the epilogues as 4 each.
    const int lcol = lane & 15, g = lane >> 4;
    unsigned goff[5][2];                      // t1 fragment offsets inside a ring row, pair 2 * wave (the odd pair: ^ 16)
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
        int tap = 2 * s5 + (g >> 1);
        if (tap > 8) tap = 8;                 // zero-weight slot: any valid location
        const int kx = tap % 3;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            int p = tt * 16 + lcol + kx - 1;
            p = p < 0 ? 0 : (p > 31 ? 31 : p);   // strip pixels 0 and 31 are halo: their outputs are not used
            goff[s5][tt] = p * 512 + (((wave * 2 + 16 * (g & 1)) ^ p) << 4);
        }
    }
    const bool k1_row1 = (g >> 1) != 0;       // k step 1: tap 3 (ky = 1) for the upper half of the lanes, tap 2 (ky = 0) below
    // grouped-conv filter fragments (mfma_16x16x32: row = out channel lane & 15 of a 16-channel pair, k = (lane >> 4) * 8 + j;
    // block diagonal over the pair's two groups, two taps per k step, zero in the 10th tap slot): 2 pairs x 5 steps, masked
    // once and RESIDENT (40 VGPRs -- affordable since the offsets above stopped being hoisted).  Re-read from LDS per row they
    // were a third of the grouped conv's LDS traffic, the stage's limit with all eight waves in it at once.
    h8_t ga[2][5];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const int tap = 2 * s5 + (g >> 1);
            const bool live = tap <= 8 && (lcol >> 3) == (g & 1);
            h8_t v = *reinterpret_cast<const h8_t *>(wg + ((((size_t)wave * 2 + pi) * 16 + lcol) * 9 + (tap > 8 ? 8 : tap)) * 8);
            if (!live) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)0.0f;
            }
            ga[pi][s5] = v;
        }
    const unsigned ssb = (unsigned)(reinterpret_cast<unsigned char *>(SS) - smem) + (wave * 32 + 4 * lhi) * 4;   // scale / shift: + immediates
    const unsigned ssg = (unsigned)(reinterpret_cast<unsigned char *>(SS) - smem) + (wave * 32 + g * 4) * 4;     // (grouped conv's lane layout)
    for (int r = ya - 1; r <= yb; ++r) {
        // ONE block barrier per row (after the grouped conv).  It publishes (1) the t2 row, (2) the x row r + 1 this wave
        // requested at the top of the iteration (its vmcnt(0) sits in front of the barrier, ~4000 cycles after the request).
        // What a second barrier at the top of the row used to protect is covered by one more buffer each: t2 rows alternate
        // between two buffers (a wave can only reach the next write of a buffer through a barrier every wave arrives at after
        // its last read of it), and the x ring has a fourth slot (the copy of row r + 1 overwrites row r - 3, whose last
        // reader finished before the previous barrier).  Without the second barrier the waves of a SIMD drift apart by up to
        // a row, and one wave's MFMA stages overlap the other's epilogues.
        RB_STAMP(0)
        if (r + 1 <= yb) { RB_ISSUE_X(r + 1) }

        // ---- conv1(r): t1 row r, this wave's 32 channels
        {
            const unsigned char *xr = XR + ((r + RB_NX) % RB_NX) * RB_PROW + bfrag;
            f32x16_t acc0;      // one accumulator chain: the SIMD's other wave fills the dependent-issue gaps
#pragma unroll
            for (int i = 0; i < 16; ++i) acc0[i] = 0.0f;
            // the next k slice's fragment is requested before the current MFMA issues (LDS returns in order, so the
            // compiler can wait with a counted lgkmcnt): LDS latency overlaps the matrix pipe inside one wave
            // four k-slice fragments in flight ahead of the MFMA that needs them (LDS returns in order: the waits are
            // counted lgkmcnt); the issue order is pinned -- left alone hipcc reads two, waits for both, issues two MFMAs
            h8_t bq[RB_PF];
#pragma unroll
            for (int j = 0; j < RB_PF; ++j) bq[j] = *reinterpret_cast<const h8_t *>(xr + j * 32);
            __builtin_amdgcn_sched_group_barrier(0x100, RB_PF, 0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kk], bq[kk % RB_PF], acc0, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kk + RB_PF < 16) {
                    bq[kk % RB_PF] = *reinterpret_cast<const h8_t *>(xr + (kk + RB_PF) * 32);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#ifdef SFD2_RB_TRACE
            asm volatile("s_nop 0" ::"v"(acc0[0]));
#endif
            RB_STAMP(1)
            const bool inside = r >= 0 && r < H && ncol >= 0 && ncol < W;
            int nv = n;
            unsigned ssv = ssb;
            asm volatile("" : "+v"(nv), "+v"(ssv));   // (recomputed per row instead of hoisted)
            unsigned char *t1w = T1 + ((r + 3) % 3) * RB_ROW + nv * 512;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = wave * 32 + 8 * q + 4 * lhi;
                const float4 s = rb_lds4o(smem, ssv + 32 * q);
                const float4 h = rb_lds4o(smem, ssv + 1024 + 32 * q);
                h4_t v = rb_cvt4(0.f, 0.f, 0.f, 0.f);
                if (inside)
                    v = rb_cvt4(fmaxf(acc0[4 * q + 0] * s.x + h.x, 0.0f), fmaxf(acc0[4 * q + 1] * s.y + h.y, 0.0f),
                                fmaxf(acc0[4 * q + 2] * s.z + h.z, 0.0f), fmaxf(acc0[4 * q + 3] * s.w + h.w, 0.0f));
                // t1 slot of channel c: (c >> 4) + 16 * ((c >> 3) & 1) -- the two 8-channel groups of a pair sit 256 B apart, so
                // the 16 lanes of a ds_read_b128 lane group of the grouped conv (two groups x 8 pixels) hit 16 distinct banks
                *reinterpret_cast<h4_t *>(t1w + ((((c0 >> 4) + 16 * ((c0 >> 3) & 1)) ^ nv) << 4) + (c0 & 4) * 2) = v;
            }
        }
        const int y = r - 1;                  // the output row this iteration finishes
        RB_STAMP(2)
        // ---- gconv(y): t2 row from t1 rows y-1, y, y+1 (this wave's own channels: no barrier needed).  Ten k steps
        // (2 channel pairs x 5 tap pairs), each two t1 fragments + two MFMAs against a resident filter fragment; the
        // fragments of step st + 1 are requested before the MFMAs of step st issue.
        if (y >= ya) {
            const unsigned t1b = (unsigned)(T1 - smem);
            const unsigned rw0 = t1b + ((y + 2) % 3) * RB_ROW, rw1 = t1b + (y % 3) * RB_ROW, rw2 = t1b + ((y + 1) % 3) * RB_ROW;
            unsigned x16 = 16;
            asm volatile("" : "+v"(x16));                 // (the odd pair's offsets: not hoisted)
            h8_t fb[2][2];
            f32x4_t acc[2][2];
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[pi][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#define RB_GLOAD(st_)                                                                                           \
            {                                                                                                   \
                const int pi_ = (st_) / 5, s_ = (st_) % 5;                                                      \
                const unsigned rw_ = (s_ == 0) ? rw0 : (s_ == 1) ? (k1_row1 ? rw1 : rw0) : (s_ == 2) ? rw1 : rw2; \
                _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                \
                    fb[(st_) & 1][tt] = *reinterpret_cast<const h8_t *>(smem + rw_ + (pi_ ? (goff[s_][tt] ^ x16) : goff[s_][tt])); \
            }
            RB_GLOAD(0)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int st = 0; st < 10; ++st) {
                if (st + 1 < 10) RB_GLOAD(st + 1)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    acc[st / 5][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga[st / 5][st % 5], fb[st & 1][tt], acc[st / 5][tt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
#undef RB_GLOAD
            unsigned sgv = ssg;
            asm volatile("" : "+v"(sgv));
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const int c0 = (wave * 2 + pi) * 16 + g * 4;
                const float4 s2 = rb_lds4o(smem, sgv + 2048 + pi * 64);
                const float4 h2 = rb_lds4o(smem, sgv + 3072 + pi * 64);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int j = tt * 16 + lcol;
                    const h4_t v = rb_cvt4(fmaxf(acc[pi][tt][0] * s2.x + h2.x, 0.0f), fmaxf(acc[pi][tt][1] * s2.y + h2.y, 0.0f),
                                           fmaxf(acc[pi][tt][2] * s2.z + h2.z, 0.0f), fmaxf(acc[pi][tt][3] * s2.w + h2.w, 0.0f));
                    *reinterpret_cast<h4_t *>(T2 + (y & 1) * RB_PROW + (j >> 1) * 1056 + (j & 1) * 512 + ((((c0 >> 3) ^ (j & 1))) << 4) + (c0 & 4) * 2) = v;
                }
            }
        }
        RB_STAMP(3)
        RB_WAIT_ALL();                        // t2 row complete (all 256 channels), x row r + 1 landed
        RB_STAMP(4)

        // ---- conv3(y) + bn3 + residual (x row y, from the LDS ring) + ReLU -> HBM
        if (y >= ya) {
            const unsigned char *tp = T2 + (y & 1) * RB_PROW + bfrag;
            f32x16_t acc0;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc0[i] = 0.0f;
            h8_t bq[RB_PF];
#pragma unroll
            for (int j = 0; j < RB_PF; ++j) bq[j] = *reinterpret_cast<const h8_t *>(tp + j * 32);
            __builtin_amdgcn_sched_group_barrier(0x100, RB_PF, 0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[kk], bq[kk % RB_PF], acc0, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (kk + RB_PF < 16) {
                    bq[kk % RB_PF] = *reinterpret_cast<const h8_t *>(tp + (kk + RB_PF) * 32);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#ifdef SFD2_RB_TRACE
            asm volatile("s_nop 0" ::"v"(acc0[0]));
#endif
            RB_STAMP(5)
            const unsigned char *xres = XR + ((y + RB_NX) % RB_NX) * RB_PROW + pp_base;
            const bool st_ok = n >= 1 && n <= RB_SW && ncol < W;
            half_t *orow = out + ((size_t)y * W + (st_ok ? ncol : 0)) * 256;
            const int cl = wave * 32 + 4 * lhi;
            unsigned ssv3 = ssb;
            asm volatile("" : "+v"(ssv3));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uint2 pk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const int c0 = cl + 8 * q;
                    const float4 s3 = rb_lds4o(smem, ssv3 + 4096 + 32 * q);
                    const float4 h3 = rb_lds4o(smem, ssv3 + 5120 + 32 * q);
                    const h4_t rs = *reinterpret_cast<const h4_t *>(xres + (((c0 >> 3) ^ (n & 1)) << 4) + (c0 & 4) * 2);
                    const h4_t hv = rb_cvt4(fmaxf(acc0[4 * q + 0] * s3.x + h3.x + (float)rs[0], 0.0f),
                                            fmaxf(acc0[4 * q + 1] * s3.y + h3.y + (float)rs[1], 0.0f),
                                            fmaxf(acc0[4 * q + 2] * s3.z + h3.z + (float)rs[2], 0.0f),
                                            fmaxf(acc0[4 * q + 3] * s3.w + h3.w + (float)rs[3], 0.0f));
                    __builtin_memcpy(&pk[j], &hv, 8);
                }
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1s = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                if (st_ok)
                    *reinterpret_cast<uint4 *>(orow + wave * 32 + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1s[0], t0[1], t1s[1]);
            }
        }
        RB_STAMP(6)
    }
#undef RB_ISSUE_X
#undef RB_WAIT_ALL
}

void launch_resblock(hipStream_t st, const half_t *x, int H, int W, const half_t *w1, const float *sc1, const float *sh1,
                     const half_t *wg, const float *sc2, const float *sh2, const half_t *w3, const float *sc3, const float *sh3,
                     half_t *out, const half_t *zero_page)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = (size_t)(RB_NX + 2) * RB_PROW + 3 * RB_ROW + 6 * 256 * sizeof(float);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(resblock_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int strips = (W + RB_SW - 1) / RB_SW;
    int segs = slots / strips;                 // one item per CU: every item pays two halo rows
    if (segs < 1) segs = 1;
    if (segs > H) segs = H;
    const int rpi = (H + segs - 1) / segs;
    segs = (H + rpi - 1) / rpi;
    hipLaunchKernelGGL(resblock_kernel, dim3(strips * segs), dim3(RB_NT), lds, st, x, H, W, w1, sc1, sh1, wg, sc2, sh2, w3,
                       sc3, sh3, out, strips, rpi, zero_page);
#ifdef SFD2_RB_TRACE
    {
        static int dumps = 0;
        if (H >= 200 && ++dumps == 40) {
            (void)hipStreamSynchronize(st);
            static unsigned long long h[2][64][8];
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rb_trace), sizeof(h));
            fprintf(stderr, "rbtrace columns: copies+conv1 MFMAs | conv1 epilogue | grouped conv | barrier | conv3 MFMAs | conv3 epilogue+store | to next row\n");
            for (int w = 0; w < 2; ++w)
                for (int r = 2; r < 10; ++r) {
                    fprintf(stderr, "rbtrace wave %d row %2d:", w * 4, r);
                    for (int k = 1; k < 7; ++k) fprintf(stderr, " %6lld", (long long)(h[w][r][k] - h[w][r][k - 1]));
                    fprintf(stderr, "  | %lld\n", (long long)(h[w][r + 1][0] - h[w][r][6]));
                }
        }
    }
#endif
}
