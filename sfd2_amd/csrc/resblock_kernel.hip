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

#define RB_NT 512
#define RB_SW 30                    // output columns per strip (32 with the halo = one MFMA pixel tile)
#ifndef RB_PF
#define RB_PF 2                     // k-slice fragments in flight ahead of the 1x1 convolutions' MFMAs
#endif
#define RB_NX 3                     // x rows in the ring (rows r - 1, r and the one in flight)
#define RB_GW_BYTES (256 * 9 * 8 * 2) // grouped-conv filters, compact [oc][tap][8 in] fp16 (36 KB)
#define RB_ROW (32 * 512)           // bytes of one 32-pixel row of 256 fp16 channels (t1 ring: XOR-swizzled records)
#define RB_PROW (16 * 1056)         // x ring / t2 row: pixel PAIRS of 1024 B + 32 B pad, slot bit 0 XORed with the pixel's parity

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ h4_t rb_cvt4(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// scale / shift of 4 consecutive channels starting at base + 4 * lhi, base wave-uniform: scalar loads (scalar cache,
// no LDS bandwidth -- the kernel is LDS-read bound) of both halves, selected per lane half
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
    unsigned char *T2 = XR + RB_NX * RB_PROW;               // one pair-padded row
    unsigned char *T1 = T2 + RB_PROW;                       // [3][32][512]
    unsigned char *GW = T1 + 3 * RB_ROW;                    // grouped-conv filters [256][9][8] fp16
    float *SS = reinterpret_cast<float *>(GW + RB_GW_BYTES);   // sc1 sh1 sc2 sh2 sc3 sh3, 256 each

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
    // grouped-conv filters stay in LDS in compact form (the block-diagonal MFMA fragments would be 80 VGPRs per wave
    // on top of the 128 of W1 / W3: that spilled, and every spill reload drains the in-flight row copies)
    for (int t = tid; t < RB_GW_BYTES / 16; t += RB_NT)
        reinterpret_cast<uint4 *>(GW)[t] = reinterpret_cast<const uint4 *>(wg)[t];
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
#define RB_WAIT_KEEP2() asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RB_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RB_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    RB_ISSUE_X(ya - 1)
    SFD2_BARRIER_DRAIN();    // GW and the first row complete

    const int n = lrow;                       // this lane's strip pixel in the 32x32 MFMAs
    const int pp_base = (n >> 1) * 1056 + (n & 1) * 512;               // its record in a pair-padded row
    const int bfrag = pp_base + ((lhi ^ (n & 1)) << 4);                // + 32 * kk = its B fragment of k slice kk
    const int ncol = col0 + n;
    for (int r = ya - 1; r <= yb; ++r) {
        // ---- x row r has landed and every wave is past conv3(r - 2); its slot takes row r + 1.  Vector-memory
        // operations retire in order: once at most the two newest (the output stores of row r - 2) are outstanding,
        // the copies of row r, issued before them, are complete.
        if (r != ya - 1) { if (r - 2 >= ya) RB_WAIT_KEEP2(); else RB_WAIT_ALL(); }   // no stores behind the first rows' copies
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
            const bool inside = r >= 0 && r < H && ncol >= 0 && ncol < W;
            unsigned char *t1w = T1 + ((r + 3) % 3) * RB_ROW + n * 512;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = wave * 32 + 8 * q + 4 * lhi;
                const float4 s = rb_lds4(SS + c0);
                const float4 h = rb_lds4(SS + 256 + c0);
                h4_t v = rb_cvt4(0.f, 0.f, 0.f, 0.f);
                if (inside)
                    v = rb_cvt4(fmaxf(acc0[4 * q + 0] * s.x + h.x, 0.0f), fmaxf(acc0[4 * q + 1] * s.y + h.y, 0.0f),
                                fmaxf(acc0[4 * q + 2] * s.z + h.z, 0.0f), fmaxf(acc0[4 * q + 3] * s.w + h.w, 0.0f));
                // t1 slot of channel c: (c >> 4) + 16 * ((c >> 3) & 1) -- the two 8-channel groups of a pair sit 256 B apart, so
                // the 16 lanes of a ds_read_b128 lane group of the grouped conv (two groups x 8 pixels) hit 16 distinct banks
                *reinterpret_cast<h4_t *>(t1w + ((((c0 >> 4) + 16 * ((c0 >> 3) & 1)) ^ n) << 4) + (c0 & 4) * 2) = v;
            }
        }
        const int y = r - 1;                  // the output row this iteration finishes
        if (y < ya) continue;

        // ---- gconv(y): t2 row from t1 rows y-1, y, y+1 (this wave's own channels: no barrier needed)
        {
            const int lcol = lane & 15, g = lane >> 4;
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const int P = wave * 2 + pi;              // 16-channel pair
                f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    int tap = 2 * s + (g >> 1);
                    if (tap > 8) tap = 8;                 // zero-weight slot: any valid location
                    const int ky = tap / 3, kx = tap - ky * 3;
                    const unsigned char *trow = T1 + ((y + ky - 1 + 3) % 3) * RB_ROW;
                    // A fragment of mfma_16x16x32 (row = out channel lane & 15 of the pair, k = (lane >> 4) * 8 + j): block
                    // diagonal over the pair's two groups, zero in the 10th tap slot
                    h8_t af = *reinterpret_cast<const h8_t *>(GW + ((P * 16 + lcol) * 9 + tap) * 16);
                    if (((lcol >> 3) != (g & 1)) || (2 * s + (g >> 1) > 8)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) af[e] = (half_t)0.0f;
                    }
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        int p = tt * 16 + lcol + kx - 1;
                        p = p < 0 ? 0 : (p > 31 ? 31 : p);   // strip pixels 0 and 31 are halo: their outputs are not used
                        const h8_t b = *reinterpret_cast<const h8_t *>(trow + p * 512 + (((P + 16 * (g & 1)) ^ p) << 4));
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, b, acc[tt], 0, 0, 0);
                    }
                }
                const int c0 = P * 16 + g * 4;
                const float4 s2 = rb_lds4(SS + 512 + c0);
                const float4 h2 = rb_lds4(SS + 768 + c0);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int j = tt * 16 + lcol;
                    const h4_t v = rb_cvt4(fmaxf(acc[tt][0] * s2.x + h2.x, 0.0f), fmaxf(acc[tt][1] * s2.y + h2.y, 0.0f),
                                           fmaxf(acc[tt][2] * s2.z + h2.z, 0.0f), fmaxf(acc[tt][3] * s2.w + h2.w, 0.0f));
                    *reinterpret_cast<h4_t *>(T2 + (j >> 1) * 1056 + (j & 1) * 512 + ((((c0 >> 3) ^ (j & 1))) << 4) + (c0 & 4) * 2) = v;
                }
            }
        }
        RB_LDS_BARRIER();                     // t2 row complete (all 256 channels)

        // ---- conv3(y) + bn3 + residual (x row y, from the LDS ring) + ReLU -> HBM
        {
            const unsigned char *tp = T2 + bfrag;
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
            const unsigned char *xres = XR + ((y + RB_NX) % RB_NX) * RB_PROW + pp_base;
            const bool st_ok = n >= 1 && n <= RB_SW && ncol < W;
            half_t *orow = out + ((size_t)y * W + (st_ok ? ncol : 0)) * 256;
            const int cl = wave * 32 + 4 * lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uint2 pk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const int c0 = cl + 8 * q;
                    const float4 s3 = rb_lds4(SS + 1024 + c0);
                    const float4 h3 = rb_lds4(SS + 1280 + c0);
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
    }
#undef RB_ISSUE_X
#undef RB_WAIT_KEEP2
#undef RB_WAIT_ALL
#undef RB_LDS_BARRIER
}

void launch_resblock(hipStream_t st, const half_t *x, int H, int W, const half_t *w1, const float *sc1, const float *sh1,
                     const half_t *wg, const float *sc2, const float *sh2, const half_t *w3, const float *sc3, const float *sh3,
                     half_t *out, const half_t *zero_page)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = (size_t)(RB_NX + 1) * RB_PROW + 3 * RB_ROW + RB_GW_BYTES + 6 * 256 * sizeof(float);
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
}
