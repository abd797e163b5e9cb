// conv1x1_c256: the ResBlock 1x1 convolutions (256 -> 256, nets/sfd2.py:30-35,41-53: conv1 + bn1 + ReLU and
// conv3 + bn3 + residual + ReLU) as a persistent streaming GEMM.
//
// These layers move 123-184 MB for 15.7 GFLOP: they are bound by how the bytes are moved, not by the MFMAs.  The
// generic implicit-GEMM kernel re-stages the 128 KB filter matrix for every 256-pixel tile, visits every pixel
// record four times (128 of its 512 bytes per K step) and pays a prologue / epilogue per tile (ablations: with
// the input loads, the output stores or the MFMAs removed it still takes 27-30 of its 34 us).  Here instead:
//   * the FILTERS LIVE IN REGISTERS for the whole kernel: wave w owns 64 output channels of one 32-pixel half,
//     its 64 x 256 A fragments are 128 VGPRs loaded once;
//   * pixels stream through a 4-stage LDS ring of 64-pixel groups (32 KB each, one contiguous piece of the NHWC
//     tensor) filled by direct-to-LDS copies three groups ahead -- 96 KB per CU in flight, no filter traffic;
//   * barriers wait with a counted vmcnt, so neither the prefetches nor the output stores are drained;
//   * every B fragment read (ds_read_b128) feeds two MFMAs; records are 512 B, 16-byte slots XOR-swizzled with
//     (pixel & 31) on the copy's source address and on the read.
#include "sfd2_internal.h"
#include <stdlib.h>
#include <algorithm>

#define NT1 512
#define GPX 64                      // pixels per group (stage)
#define NST 4                       // stages in the ring
#define STAGE_BYTES (GPX * 512)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ h4_t c1_cvt4(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

template <bool HAS_RES>
__global__ __launch_bounds__(NT1, 2)
void conv1x1_c256_kernel(const half_t *__restrict__ in, int npix, const half_t *__restrict__ w /*[256 out][256 in]*/,
                         const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                         const half_t *__restrict__ res, half_t *__restrict__ out, int groups_per_block,
                         const half_t *__restrict__ zero_page /* >= 512 B of zeros */)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                                              // [NST][GPX][512 B]
    float *SS = reinterpret_cast<float *>(smem + NST * STAGE_BYTES);       // scale[256], shift[256]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int cg = wave & 3, ph = wave >> 2;       // 64-channel group, 32-pixel half of the stage
    const int ngroups = (npix + GPX - 1) / GPX;
    const int g0 = blockIdx.x * groups_per_block;
    int g1 = g0 + groups_per_block;
    if (g1 > ngroups) g1 = ngroups;
    if (g0 >= g1) return;

    // filters of this wave: A fragments of mfma_32x32x16 (row = channel, 8 consecutive k per lane)
    h8_t a[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            a[t][kk] = *reinterpret_cast<const h8_t *>(w + (size_t)(cg * 64 + t * 32 + lrow) * 256 + kk * 16 + lhi * 8);
    for (int t = tid; t < 256; t += NT1) { SS[t] = scale[t]; SS[256 + t] = shift[t]; }

    // group g -> ring stage: 32 one-KB chunks (2 pixel records each), 4 per wave
#define ISSUE_G(g_)                                                                                        \
    {                                                                                                      \
        unsigned char *st = Xs + ((g_) & (NST - 1)) * STAGE_BYTES;                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                    \
            const int ch = wave * 4 + i;                                                                   \
            const int p = ch * 2 + lhi;                            /* pixel within the stage */            \
            const long long gp = (long long)(g_)*GPX + p;                                                  \
            const half_t *src = gp < npix ? in + gp * 256 + ((lrow ^ (p & 31)) << 3) : zero_page + (lrow << 3); \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src, (lds_void_t *)(st + ch * 1024), 16, 0, 0); \
        }                                                                                                  \
    }
    // Counted wait: loads retire in order, so group g's copies have landed once no more than N vector-memory
    // operations are outstanding, N = the LOADS that can still be in flight behind them when iteration g starts: the
    // copies of g+1 and g+2 (8).  Nothing else is counted on: an iteration's residual loads were consumed by its own
    // epilogue, and stores may retire early -- with them (or with the residual loads, as until round 4: 12) in N, a
    // wave whose stores had drained could pass with group g's copies still pending.  A smaller N only waits longer.
#define WAIT_GROUP()                                                                                       \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    ISSUE_G(g0)
    if (g0 + 1 < g1) { ISSUE_G(g0 + 1) }
    if (g0 + 2 < g1) { ISSUE_G(g0 + 2) }
    SFD2_BARRIER_DRAIN();   // first group, SS (and everything else) complete

    for (int g = g0; g < g1; ++g) {
        if (g != g0) {
            // tail: fewer copies are in flight than the constant assumes -> drain (at most twice per block)
            if (g + 2 < g1) WAIT_GROUP(); else SFD2_BARRIER_DRAIN();
        }
        const unsigned char *st = Xs + (g & (NST - 1)) * STAGE_BYTES;
        const int p = ph * 32 + lrow;            // this lane's pixel within the stage
        const long long gp = (long long)g * GPX + p;
        const bool inb = gp < npix;
        const size_t obase = (size_t)(inb ? gp : 0) * 256 + cg * 64;

        uint4 rq[2][2];
        if (HAS_RES) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    rq[t][m] = make_uint4(0, 0, 0, 0);
                    if (inb) rq[t][m] = *reinterpret_cast<const uint4 *>(res + obase + t * 32 + 8 * (2 * m + lhi));
                }
            asm volatile("" ::: "memory");       // keep the residual loads ahead of the copies below in program order
        }
        if (g + 3 < g1) { ISSUE_G(g + 3) }      // into the stage group g - 1 used: every wave is past it (barrier above)

        f32x16_t acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const unsigned char *xp = st + p * 512;
        const int sw = p & 31;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][kk], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][kk], b, acc[1], 0, 0, 0);
        }

        // epilogue: y = acc * scale + shift (+ residual) (ReLU); v_permlane32_swap regroups two channel quads so each
        // lane stores (and loaded its residual as) 16-byte runs -- same scheme as conv_igemm2
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int cl = cg * 64 + t * 32 + 4 * lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uint2 rp[2] = {make_uint2(0, 0), make_uint2(0, 0)};
                if (HAS_RES) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(rq[t][m].x, rq[t][m].z, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(rq[t][m].y, rq[t][m].w, false, false);
                    rp[0] = make_uint2(s0[0], s1[0]);
                    rp[1] = make_uint2(s0[1], s1[1]);
                }
                uint2 pk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const float4 sc = sfd2_lds_f4(SS + cl + 8 * q);          // (typed like the fragments: no vmcnt(0) drain of the ring)
                    const float4 sh = sfd2_lds_f4(SS + 256 + cl + 8 * q);
                    float v0 = acc[t][4 * q + 0] * sc.x + sh.x;
                    float v1 = acc[t][4 * q + 1] * sc.y + sh.y;
                    float v2 = acc[t][4 * q + 2] * sc.z + sh.z;
                    float v3 = acc[t][4 * q + 3] * sc.w + sh.w;
                    if (HAS_RES) {
                        h4_t r;
                        __builtin_memcpy(&r, &rp[j], 8);
                        v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
                    }
                    if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
                    const h4_t hv = c1_cvt4(v0, v1, v2, v3);
                    __builtin_memcpy(&pk[j], &hv, 8);
                }
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                if (inb)
                    *reinterpret_cast<uint4 *>(out + obase + t * 32 + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            }
        }
    }
#undef ISSUE_G
#undef WAIT_GROUP
}

void launch_conv1x1_c256(hipStream_t st, const half_t *in, int npix, const half_t *w_rowmajor, const float *scale,
                         const float *shift, int relu, const half_t *res, half_t *out, const half_t *zero_page)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = (size_t)NST * STAGE_BYTES + 512 * sizeof(float);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_c256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_c256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;   // 130 KB of LDS, 8 waves of <= 256 VGPRs: one resident block per CU
        attr_done = true;
    }
    const int ngroups = (npix + GPX - 1) / GPX;
    if (ngroups == 0) return;
    const int gpb = (ngroups + sfd2_slots(slots) - 1) / sfd2_slots(slots);
    const int grid = (ngroups + gpb - 1) / gpb;
    if (res) hipLaunchKernelGGL(conv1x1_c256_kernel<true>, dim3(grid), dim3(NT1), lds, st, in, npix, w_rowmajor, scale, shift, relu, res, out, gpb, zero_page);
    else hipLaunchKernelGGL(conv1x1_c256_kernel<false>, dim3(grid), dim3(NT1), lds, st, in, npix, w_rowmajor, scale, shift, relu, res, out, gpb, zero_page);
}

// ---------------------------------------------------------------------------------------------
// The same layers compensated (SFD2_PREC_F16C, sfd2_internal.h): activations are a hi plane and a corr plane, filters the
// fp16 matrix and a matrix of corr units.  Same streaming structure; what changes is the split: a wave owns 32 output
// channels (64 registers of fp16 fragments + 64 of corr fragments, the same 128 as the fp16 kernel's 64 channels) and ALL
// 32 pixels of a group; a group = 32 pixels = 16 KB of hi + 16 KB of corr records, so the four-stage ring, the four copies
// per wave and group and the counted waits are those of the fp16 kernel.  Per group and wave: 16 fp16 MFMAs + 8 fp8 MFMAs.
#define GPXC 32
#define STAGE_C (2 * GPXC * 512)

// Blocks take a static contiguous share of the groups.  A wall-clock trace (-DSFD2_C256_TRACE) shows the blocks of one launch
// finishing between 55 and 70 us for identical work (mean 64), so handing groups out dynamically (an atomic claim counter,
// claims five ahead through an LDS ring) was tried: the blocks then finish within 2 us of each other, but LATER (mean 73) --
// with one counter or with eight -- and the launch is slower (conv1 47 -> 66 us, conv3 78 -> 84): not kept.
// IN_C = false: the input is a plain fp16 tensor (ResBlock-internal tensors under option "rb_inner", api_network.hip): no corr
// plane to stage, and the filter residuals arrive as fp16 (w - fp16(w)) * 2^11 (`wc` = [256][256] halves) for a second fp16
// pass into its own accumulator.  OUT_C = false: only the hi plane is written.
// IN_C = 2 (option "trunk_r1"): the input's corr plane holds ONE byte per channel, the residual e4m3((x - hi) * 2^9), 256 bytes per pixel.
// The value byte of a unit is rebuilt from the hi plane in registers -- a lane's fp16 fragments of K steps 2 c and 2 c + 1 ARE the 16
// channels whose units its corr fragment of chunk c holds: v_cvt_scalef32_pk_fp8_f16 (two values per instruction) and a byte permute per
// two channels -- so filters, MFMAs and results' structure are those of IN_C = 1, with 24 KB staged per group instead of 32.
#ifndef SFD2_R1_VSCALE
#define SFD2_R1_VSCALE 4.0f
#endif
// Ring depth of the residual-byte form (IN_C = 2, the default path's ResBlock.conv1): SFD2_C256_R1_STAGES stages of exactly 24 KB (16 KB of
// hi records + 8 KB of residual bytes), filled (stages - 1) groups ahead.  Round 5 asked whether this kernel's 3.4 TB/s is the bytes it keeps in
// flight (the fp16 kernel holds 3 x 32 KB ahead and reaches 4.0-4.4 TB/s; this form 3 x 24 KB): six stages = 120 KB in flight in 146 KB of LDS
// measured conv1 44.5 / 43.6 / 43.4 us against 44.0 / 42.6 / 41.5 with four (same box, interleaved) -- it is not; four stages stay.
#ifndef SFD2_C256_R1_STAGES
#define SFD2_C256_R1_STAGES 4
#endif
#define STAGE_R1 (GPXC * 512 + GPXC * 256)
// Round 5, what holds ResBlock.conv1 (IN_C = 2) at 42 us = 3.7 TB/s.  Ablations (timing only): without the output stores 38 us, without the
// staging copies 38 us -- neither side of the memory traffic.  A section trace (-DSFD2_C256_TRACE) puts a group at ~4 300 cycles: wait + barrier
// 170-1 200, fragment reads + value-byte rebuild + 24 MFMAs 2 300-3 500 (two waves per SIMD: 2 080 of MFMA issue), epilogue 540-660.  Tried, each
// bit-compatible and measured on one box against the shipped form:
//   SFD2_C256_STAGGER = 1  the waves as two groups one barrier apart (section A = reads + MFMAs + copies, section B = epilogue; conv3x3_pp's
//                          schedule): 46.9-48.9 us against 42.6-43.1 -- the second barrier costs more than the overlap returns
//   SFD2_C256_SPLIT_ACC = 1 the scaled MFMAs in an accumulator chain of their own (two chains of one MFMA type per wave): 44.7 / 44.1 / 44.3 against
//                          45.0 / 44.5 / 44.2
//   SFD2_C256_R1_STAGES = 6, SFD2_C256_INTERLEAVE: see below.  All off.
#ifndef SFD2_C256_STAGGER
#define SFD2_C256_STAGGER 0
#endif
#ifndef SFD2_C256_SPLIT_ACC
#define SFD2_C256_SPLIT_ACC 0
#endif
// -DSFD2_C256_TRACE: cycle stamps of block 3's waves 0 and 4 around the sections of groups 4 .. 9 (top of the iteration, behind the wait + barrier,
// behind the fragment reads + MFMAs, behind the epilogue), printed by the launcher of the residual-byte form after 40 launches
#ifdef SFD2_C256_TRACE
#include <stdio.h>
__device__ unsigned long long g_c256_cyc[2][6][4];
#define C256_CYC(k_) if (blockIdx.x == 3 && (wave == 0 || wave == 4) && lane == 0 && g - g0 >= 4 && g - g0 < 10) g_c256_cyc[wave == 4][g - g0 - 4][k_] = __builtin_readcyclecounter();
#else
#define C256_CYC(k_)
#endif
// -DSFD2_C256_ABL (timing only, wrong results): bit 0 = no output stores, bit 1 = no staging copies inside the group loop
#ifdef SFD2_C256_ABL
#define C256_ABL SFD2_C256_ABL
#else
#define C256_ABL 0
#endif
template <bool HAS_RES, int IN_C, bool OUT_C>
__global__ __launch_bounds__(NT1, 2)
void conv1x1_c256_c_kernel(const half_t *__restrict__ in, const half_t *__restrict__ in_c, int npix,
                           const half_t *__restrict__ w /*fp16 filters*/, const half_t *__restrict__ wc /*corr units, or fp16 residuals*/,
                           /* both in fragment order [8 waves][8 c][64 lanes][16]: element e of lane l = filter row wave * 32 + (l & 31),
                              column c * 32 + (e >> 3) * 16 + (l >> 5) * 8 + (e & 7) -- every load instruction reads whole lines
                              (row-major filters cost each CU ~1 MB of 32-byte pieces through its L1: 7.6 us before the first group
                              was staged, 4.7 us now) */
                           const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                           const half_t *__restrict__ res, const half_t *__restrict__ res_c,
                           half_t *__restrict__ out, half_t *__restrict__ out_c, int groups_per_block,
                           const half_t *__restrict__ zero_page, int sa, unsigned int *__restrict__ range /* the output tensor's range-status slot */)
{
    constexpr int NSTC = IN_C == 2 ? SFD2_C256_R1_STAGES : NST;           // ring stages
    constexpr int STB = IN_C == 2 ? STAGE_R1 : STAGE_C;                    // bytes per stage
    constexpr int AHEAD = NSTC - 1;                                        // groups requested ahead of the one being computed
    constexpr int CPG = IN_C == 1 ? 4 : IN_C == 2 ? 3 : 2;                 // copies per group and wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                                              // [NSTC][hi 32 x 512 B | corr 32 x 512 B (IN_C = 2: 32 x 256 B)]
    float *SS = reinterpret_cast<float *>(smem + NSTC * STB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int ngroups = (npix + GPXC - 1) / GPXC;
    // The block's groups, as ring positions g0 .. g1 - 1 -> group numbers GRP(g).  -DSFD2_C256_INTERLEAVE (experiment): block b takes groups b,
    // b + grid, ... (at any moment the CUs work in one window of grid x 24 KB of the tensor) instead of a contiguous share of groups_per_block.
    // Measured round 5, same box, interleaved: conv1 41.8 / 40.8 / 40.7 us contiguous, 41.7 / 41.4 / 42.0 interleaved -- the order the CUs walk
    // the tensor in is not what holds this kernel at 3.5-3.7 TB/s either.
#ifdef SFD2_C256_INTERLEAVE
    const int g0 = 0;
    const int g1 = ((int)blockIdx.x < ngroups) ? (ngroups - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
#define GRP(g_) ((int)blockIdx.x + (g_) * (int)gridDim.x)
#else
    const int g0 = blockIdx.x * groups_per_block;
    int g1 = g0 + groups_per_block;
    if (g1 > ngroups) g1 = ngroups;
#define GRP(g_) (g_)
#endif
    if (g0 >= g1) return;

    h8_t ah[16];
    v8i_t ac[8];               // IN_C: corr units; otherwise the 16 fp16 fragments of the scaled filter residuals, two per entry
    {
        const size_t fo = ((size_t)wave * 8 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[2 * c] = *reinterpret_cast<const h8_t *>(w + fo + (size_t)c * 64 * 16);
            ah[2 * c + 1] = *reinterpret_cast<const h8_t *>(w + fo + (size_t)c * 64 * 16 + 8);
            ac[c] = sfd2_cat8(*reinterpret_cast<const h8_t *>(wc + fo + (size_t)c * 64 * 16), *reinterpret_cast<const h8_t *>(wc + fo + (size_t)c * 64 * 16 + 8));
        }
    }
    for (int t = tid; t < 256; t += NT1) { SS[t] = scale[t]; SS[256 + t] = shift[t]; }

    // group g -> ring stage: 16 + 16 one-KB chunks (2 pixel records each): 2 of each plane per wave
#define ISSUE_GC(g_)                                                                                       \
    {                                                                                                      \
        unsigned char *st = Xs + (unsigned)((g_) - g0) % (unsigned)NSTC * STB;                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
            const int ch = wave * 2 + i;                                                                   \
            const int p = ch * 2 + lhi;                                                                    \
            const long long gp = (long long)GRP(g_)*GPXC + p;                                                 \
            const size_t so = (size_t)gp * 256 + ((lrow ^ (p & 31)) << 3);                                 \
            const half_t *s0 = gp < npix ? in + so : zero_page + (lrow << 3);                              \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)s0, (lds_void_t *)(st + ch * 1024), 16, 0, 0);  \
            if (IN_C == 1) {                                                                               \
                const half_t *s1 = gp < npix ? in_c + so : zero_page + (lrow << 3);                        \
                __builtin_amdgcn_global_load_lds((gbl_void_t *)s1, (lds_void_t *)(st + GPXC * 512 + ch * 1024), 16, 0, 0); \
            }                                                                                              \
        }                                                                                                  \
        if (IN_C == 2) {   /* residual bytes: 256 B per pixel, ONE one-KB chunk (4 pixels) per wave, 16-byte slots XOR (pixel & 15) */ \
            const int p = wave * 4 + (lane >> 4);                                                          \
            const long long gp = (long long)GRP(g_)*GPXC + p;                                                 \
            const unsigned char *s1 = gp < npix ? reinterpret_cast<const unsigned char *>(in_c) + (size_t)gp * 256 + (((lane & 15) ^ (p & 15)) << 4) \
                                                : reinterpret_cast<const unsigned char *>(zero_page) + ((lane & 15) << 4); \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)s1, (lds_void_t *)(st + GPXC * 512 + wave * 1024), 16, 0, 0); \
        }                                                                                                  \
    }
#define WAIT_GROUP_C()                                                                                     \
    /* (N = the copies of the AHEAD - 1 younger groups, loads only: see WAIT_GROUP) */                    \
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((AHEAD - 1) * CPG) : "memory")

#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (g0 + a < g1) { ISSUE_GC(g0 + a) }
    SFD2_BARRIER_DRAIN();

    float mx = 0.0f;     // range status: the largest output value in front of the saturation
    constexpr bool STG = SFD2_C256_STAGGER != 0;
    const int grp = wave >> 2;                              // waves w and w + 4 share a SIMD
    if (STG && grp == 1) asm volatile("s_barrier" ::: "memory");
    for (int g = g0; g < g1; ++g) {
        C256_CYC(0)
        if (!STG && g != g0) {
            if (g + AHEAD - 1 < g1) WAIT_GROUP_C(); else SFD2_BARRIER_DRAIN();
        }
        C256_CYC(1)
        const unsigned char *st = Xs + (unsigned)(g - g0) % (unsigned)NSTC * STB;
        const int p = lrow;
        const long long gp = (long long)GRP(g) * GPXC + p;
        const bool inb = gp < npix;
        const size_t obase = (size_t)(inb ? gp : 0) * 256 + wave * 32;

        uint4 rq[2], rc[2];
        if (HAS_RES) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                rq[m] = make_uint4(0, 0, 0, 0);
                rc[m] = make_uint4(0, 0, 0, 0);
                if (inb) {
                    rq[m] = *reinterpret_cast<const uint4 *>(res + obase + 8 * (2 * m + lhi));
                    rc[m] = *reinterpret_cast<const uint4 *>(res_c + obase + 8 * (2 * m + lhi));
                }
            }
            asm volatile("" ::: "memory");       // keep the residual loads ahead of the copies below in program order
        }
        if (!(C256_ABL & 2) && g + AHEAD < g1) { ISSUE_GC(g + AHEAD) }

        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const unsigned char *xp = st + p * 512;
        const int sw = p & 31;
        f32x16_t acl;
        if (!IN_C) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acl[r] = 0.0f;
        }
        if (IN_C == 2) {
#if SFD2_C256_SPLIT_ACC
            f32x16_t acs;       // the scaled MFMAs' own chain: a wave then runs two independent accumulator chains of one MFMA type each
#pragma unroll
            for (int r = 0; r < 16; ++r) acs[r] = 0.0f;
#endif
            const unsigned char *xr = st + GPXC * 512 + p * 256 + 8 * lhi;
            const int sr = p & 15;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const h8_t b0 = *reinterpret_cast<const h8_t *>(xp + (((c * 4 + lhi) ^ sw) << 4));
                const h8_t b1 = *reinterpret_cast<const h8_t *>(xp + (((c * 4 + 2 + lhi) ^ sw) << 4));
                // (read as halves like the fragments: behind a differently typed LDS read the compiler drains the ring's copies in flight, vmcnt(0))
                const h4_t r0h = *reinterpret_cast<const h4_t *>(xr + (((2 * c) ^ sr) << 4)), r1h = *reinterpret_cast<const h4_t *>(xr + (((2 * c + 1) ^ sr) << 4));
                uint2 r0, r1;
                __builtin_memcpy(&r0, &r0h, 8);
                __builtin_memcpy(&r1, &r1h, 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2 * c], b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[2 * c + 1], b1, acc, 0, 0, 0);
                // units (residual byte, e4m3(hi / 4)) of this lane's 16 channels
                typedef short s2v_t __attribute__((ext_vector_type(2)));
                v8i_t bu;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const h8_t bb = h ? b1 : b0;
                    const uint2 rr = h ? r1 : r0;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {   // channels 4 q .. 4 q + 3 of the fragment
                        s2v_t v = {0, 0};
                        v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(v, h2_t{bb[4 * q], bb[4 * q + 1]}, SFD2_R1_VSCALE, false);
                        v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(v, h2_t{bb[4 * q + 2], bb[4 * q + 3]}, SFD2_R1_VSCALE, true);
                        unsigned int vb;
                        __builtin_memcpy(&vb, &v, 4);
                        // K order of this form's fragments (filters packed to match, api_weights.hip `wfr`): the lane's 16 residual bytes, then
                        // its 16 value bytes -- no byte interleave
                        bu[2 * h + q] = (int)(q ? rr.y : rr.x);
                        bu[4 + 2 * h + q] = (int)vb;
                    }
                }
#if SFD2_C256_SPLIT_ACC
                acs = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ac[c], bu, acs, 0, 0, 0, sa, 0, 0x7f7f7f7f);
#else
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ac[c], bu, acc, 0, 0, 0, sa, 0, 0x7f7f7f7f);
#endif
            }
#if SFD2_C256_SPLIT_ACC
            asm volatile("" : "+v"(acs));
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acs[r];
#endif
            asm volatile("" : "+v"(acc));
        } else
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], b, acc, 0, 0, 0);
            if (!IN_C) acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(ac[kk >> 1], kk & 1), b, acl, 0, 0, 0);
        }
        if (IN_C == 2) {
        } else if (IN_C) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const v8i_t b = sfd2_cat8(*reinterpret_cast<const h8_t *>(xp + GPXC * 512 + (((c * 4 + lhi) ^ sw) << 4)),
                                          *reinterpret_cast<const h8_t *>(xp + GPXC * 512 + (((c * 4 + 2 + lhi) ^ sw) << 4)));
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ac[c], b, acc, 0, 0, 0, sa, 0, 0x7f7f7f7f);
            }
            asm volatile("" : "+v"(acc));   // (the scaled MFMA is a pure node to instruction selection: keep it in front of the epilogue)
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(acl[r], 1.0f / 2048.0f, acc[r]);
        }

        asm volatile("" : "+v"(acc));
        C256_CYC(2)
        if (STG) {
            // end of section A: this wave's copies of group g + 1 have landed (the AHEAD - 1 younger groups may stay in flight; near the
            // tail nothing younger exists: full wait), its fragment reads are done; the barrier hands the matrix pipe to the other group
            asm volatile("" : "+v"(acc));
            if (g + AHEAD < g1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((AHEAD - 1) * CPG) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        const int cl = wave * 32 + 4 * lhi;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            uint2 rp[2] = {make_uint2(0, 0), make_uint2(0, 0)}, rcp[2] = {make_uint2(0, 0), make_uint2(0, 0)};
            if (HAS_RES) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(rq[m].x, rq[m].z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(rq[m].y, rq[m].w, false, false);
                rp[0] = make_uint2(s0[0], s1[0]);
                rp[1] = make_uint2(s0[1], s1[1]);
                const auto c0 = __builtin_amdgcn_permlane32_swap(rc[m].x, rc[m].z, false, false);
                const auto c1 = __builtin_amdgcn_permlane32_swap(rc[m].y, rc[m].w, false, false);
                rcp[0] = make_uint2(c0[0], c1[0]);
                rcp[1] = make_uint2(c0[1], c1[1]);
            }
            uint2 pk[2], ck[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = 2 * m + j;
                const float4 sc = sfd2_lds_f4(SS + cl + 8 * q);
                const float4 sh = sfd2_lds_f4(SS + 256 + cl + 8 * q);
                float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                if (HAS_RES) {
                    h4_t r;
                    __builtin_memcpy(&r, &rp[j], 8);
                    ad = make_float4((float)r[0] + sfd2_corr_lo(rcp[j].x, 0), (float)r[1] + sfd2_corr_lo(rcp[j].x, 1),
                                     (float)r[2] + sfd2_corr_lo(rcp[j].y, 0), (float)r[3] + sfd2_corr_lo(rcp[j].y, 1));
                }
                sfd2_epi4<HAS_RES>(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], sc, sh, ad, relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j], mx, inb);
            }
            const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
            const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
            if (inb && (!(C256_ABL & 1) || t0[0] == 0x12345678u)) *reinterpret_cast<uint4 *>(out + obase + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
            if (OUT_C) {
                const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                if (inb) *reinterpret_cast<uint4 *>(out_c + obase + 8 * (2 * m + lhi)) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
            }
        }
        C256_CYC(3)
        if (STG && !(grp == 1 && g + 1 == g1)) {            // end of section B (group 1's last one has nobody left to hand over to)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    sfd2_range_commit(range, sfd2_wave_max_bits(mx));
#undef ISSUE_GC
#undef GRP
#undef WAIT_GROUP_C
}

// ---------------------------------------------------------------------------------------------
// SFD2_PREC_F16X3 for the same layers: the input as hi / lo' planes (x3_split's arithmetic: hi = fp16(x), lo' = fp16((x - hi) * 2^11)),
// filters as fp16 + lo' fragments in the fragment order above, three MFMAs per K slice into two accumulators (hi x hi;
// hi x lo' + lo' x hi, weighted 2^-11 -- conv_igemm_x3_kernel's combination).  RES: 0 none, 1 fp32 residual [P][256], 2 the
// residual as hi / lo' planes (res_v / res_lo; hi + lo' * 2^-11: 22 significant bits).  OUT_F32: fp32 output
// [P][256]; OUT_PLANES: the output as planes for the layer that reads it next (may be the residual's own buffers: a lane reads its
// residual values before it stores to the same addresses).  The streaming structure is the compensated kernel's with both planes
// staged: a group = 16 KB hi + 16 KB lo'.
template <int RES, bool OUT_F32, bool OUT_PLANES>
__global__ __launch_bounds__(NT1, 2)
void conv1x1_c256_x3_kernel(const half_t *__restrict__ in, const half_t *__restrict__ in_lo, int npix,
                            const half_t *__restrict__ w, const half_t *__restrict__ wl,
                            const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                            const void *res_v, const half_t *res_lo, float *__restrict__ out, half_t *out_hi, half_t *out_lo,
                            int groups_per_block, const half_t *__restrict__ zero_page)
{
    constexpr bool HAS_RES = RES != 0;
    const float *res = static_cast<const float *>(res_v);
    const half_t *res_hi = static_cast<const half_t *>(res_v);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                                              // [NST][hi 32 x 512 B | lo' 32 x 512 B]
    float *SS = reinterpret_cast<float *>(smem + NST * STAGE_C);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    const int ngroups = (npix + GPXC - 1) / GPXC;
    const int g0 = blockIdx.x * groups_per_block;
    int g1 = g0 + groups_per_block;
    if (g1 > ngroups) g1 = ngroups;
    if (g0 >= g1) return;

    h8_t ah[16];
    v8i_t al[8];
    {
        const size_t fo = ((size_t)wave * 8 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[2 * c] = *reinterpret_cast<const h8_t *>(w + fo + (size_t)c * 64 * 16);
            ah[2 * c + 1] = *reinterpret_cast<const h8_t *>(w + fo + (size_t)c * 64 * 16 + 8);
            al[c] = sfd2_cat8(*reinterpret_cast<const h8_t *>(wl + fo + (size_t)c * 64 * 16), *reinterpret_cast<const h8_t *>(wl + fo + (size_t)c * 64 * 16 + 8));
        }
    }
    for (int t = tid; t < 256; t += NT1) { SS[t] = scale[t]; SS[256 + t] = shift[t]; }

#define ISSUE_GX(g_)                                                                                       \
    {                                                                                                      \
        unsigned char *st = Xs + ((g_) & (NST - 1)) * STAGE_C;                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
            const int ch = wave * 2 + i;                                                                   \
            const int p = ch * 2 + lhi;                                                                    \
            const long long gp = (long long)(g_)*GPXC + p;                                                 \
            const size_t so = (size_t)gp * 256 + ((lrow ^ (p & 31)) << 3);                                 \
            const half_t *s0 = gp < npix ? in + so : zero_page + (lrow << 3);                              \
            const half_t *s1 = gp < npix ? in_lo + so : zero_page + (lrow << 3);                           \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)s0, (lds_void_t *)(st + ch * 1024), 16, 0, 0);  \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)s1, (lds_void_t *)(st + GPXC * 512 + ch * 1024), 16, 0, 0); \
        }                                                                                                  \
    }
    // per group and wave: 4 copies, 4 residual loads (fp32, or 16 B of each plane per channel-run pair), 4 fp32 stores, 4 plane stores
#define WAIT_GROUP_X()                                                                                     \
    /* (N = the copies of the two younger groups, loads only: see WAIT_GROUP) */                          \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory")

    ISSUE_GX(g0)
    if (g0 + 1 < g1) { ISSUE_GX(g0 + 1) }
    if (g0 + 2 < g1) { ISSUE_GX(g0 + 2) }
    SFD2_BARRIER_DRAIN();

    for (int g = g0; g < g1; ++g) {
        if (g != g0) {
            if (g + 2 < g1) WAIT_GROUP_X(); else SFD2_BARRIER_DRAIN();
        }
        const unsigned char *st = Xs + (g & (NST - 1)) * STAGE_C;
        const int p = lrow;
        const long long gp = (long long)g * GPXC + p;
        const bool inb = gp < npix;
        const size_t obase = (size_t)(inb ? gp : 0) * 256 + wave * 32 + 4 * lhi;

        float4 rr[4];
        uint4 rh16[2], rl16[2];     // RES == 2: a lane pair's 8 + 8 channels of a run pair as one 16-byte load each (regrouped below)
        if (RES == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[q] = *reinterpret_cast<const float4 *>(res + obase + 8 * q);   // (pixels past the end: pixel 0's)
            asm volatile("" ::: "memory");       // keep the residual loads ahead of the copies below in program order
        }
        if (RES == 2) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const size_t o16 = (size_t)(inb ? gp : 0) * 256 + wave * 32 + 8 * (2 * m + lhi);
                rh16[m] = *reinterpret_cast<const uint4 *>(res_hi + o16);
                rl16[m] = *reinterpret_cast<const uint4 *>(res_lo + o16);
            }
            asm volatile("" ::: "memory");
        }
        if (g + 3 < g1) { ISSUE_GX(g + 3) }

        f32x16_t acc, acl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acl[r] = 0.0f; }
        const unsigned char *xp = st + p * 512;
        const int sw = p & 31;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
            const h8_t bl = *reinterpret_cast<const h8_t *>(xp + GPXC * 512 + (((kk * 2 + lhi) ^ sw) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], b, acc, 0, 0, 0);
            acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl, acl, 0, 0, 0);
            acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(al[kk >> 1], kk & 1), b, acl, 0, 0, 0);
        }
        const int cl = wave * 32 + 4 * lhi;
        h4_t rh[4], rl[4];
        if (RES == 2) {
            // lane (lhi = 0) loaded channels 16 m .. + 7 = run 2 m of both half-waves, lane (lhi = 1) run 2 m + 1 of both:
            // v_permlane32_swap hands each lane its own two runs (the store regrouping below, reversed)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const auto a0 = __builtin_amdgcn_permlane32_swap(rh16[m].x, rh16[m].z, false, false);
                const auto a1 = __builtin_amdgcn_permlane32_swap(rh16[m].y, rh16[m].w, false, false);
                const auto b0 = __builtin_amdgcn_permlane32_swap(rl16[m].x, rl16[m].z, false, false);
                const auto b1 = __builtin_amdgcn_permlane32_swap(rl16[m].y, rl16[m].w, false, false);
                const uint2 h0 = make_uint2(a0[0], a1[0]), h1 = make_uint2(a0[1], a1[1]), l0 = make_uint2(b0[0], b1[0]), l1 = make_uint2(b0[1], b1[1]);
                __builtin_memcpy(&rh[2 * m], &h0, 8); __builtin_memcpy(&rh[2 * m + 1], &h1, 8);
                __builtin_memcpy(&rl[2 * m], &l0, 8); __builtin_memcpy(&rl[2 * m + 1], &l1, 8);
            }
        }
        uint2 ph[4], pl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sc = sfd2_lds_f4(SS + cl + 8 * q);
            const float4 sh = sfd2_lds_f4(SS + 256 + cl + 8 * q);
            float v0 = (acc[4 * q + 0] + acl[4 * q + 0] * (1.0f / 2048.0f)) * sc.x + sh.x;
            float v1 = (acc[4 * q + 1] + acl[4 * q + 1] * (1.0f / 2048.0f)) * sc.y + sh.y;
            float v2 = (acc[4 * q + 2] + acl[4 * q + 2] * (1.0f / 2048.0f)) * sc.z + sh.z;
            float v3 = (acc[4 * q + 3] + acl[4 * q + 3] * (1.0f / 2048.0f)) * sc.w + sh.w;
            if (RES == 2) {
                rr[q] = make_float4((float)rh[q][0] + (float)rl[q][0] * (1.0f / 2048.0f), (float)rh[q][1] + (float)rl[q][1] * (1.0f / 2048.0f),
                                    (float)rh[q][2] + (float)rl[q][2] * (1.0f / 2048.0f), (float)rh[q][3] + (float)rl[q][3] * (1.0f / 2048.0f));
            }
            if (HAS_RES) { v0 += rr[q].x; v1 += rr[q].y; v2 += rr[q].z; v3 += rr[q].w; }
            if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
            if (OUT_F32 && inb) *reinterpret_cast<float4 *>(out + obase + 8 * q) = make_float4(v0, v1, v2, v3);
            if (OUT_PLANES) {
                h4_t hv = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                h4_t lv = {(half_t)((v0 - (float)hv[0]) * 2048.0f), (half_t)((v1 - (float)hv[1]) * 2048.0f),
                           (half_t)((v2 - (float)hv[2]) * 2048.0f), (half_t)((v3 - (float)hv[3]) * 2048.0f)};
                __builtin_memcpy(&ph[q], &hv, 8);
                __builtin_memcpy(&pl[q], &lv, 8);
            }
        }
        if (OUT_PLANES) {
            // 16-byte plane stores: runs 2 m / 2 m + 1 of the two half-waves regrouped with v_permlane32_swap (conv2_kernels.hip)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const auto t0 = __builtin_amdgcn_permlane32_swap(ph[2 * m].x, ph[2 * m + 1].x, false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(ph[2 * m].y, ph[2 * m + 1].y, false, false);
                const auto u0 = __builtin_amdgcn_permlane32_swap(pl[2 * m].x, pl[2 * m + 1].x, false, false);
                const auto u1 = __builtin_amdgcn_permlane32_swap(pl[2 * m].y, pl[2 * m + 1].y, false, false);
                const size_t o16 = (size_t)(inb ? gp : 0) * 256 + wave * 32 + 8 * (2 * m + lhi);
                if (inb) {
                    *reinterpret_cast<uint4 *>(out_hi + o16) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                    *reinterpret_cast<uint4 *>(out_lo + o16) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                }
            }
        }
    }
#undef ISSUE_GX
#undef WAIT_GROUP_X
}

// in / in_lo: input planes; w / wl: fp16 filters and their lo' parts in fragment order; res: null, fp32 [P][256] (res_lo null) or the hi
// plane with res_lo its lo' plane; out: fp32 [P][256] or null; out_hi / out_lo: the output as planes, or null (one of the two forms)
void launch_conv1x1_c256_x3(hipStream_t st, const half_t *in, const half_t *in_lo, int npix, const half_t *w, const half_t *wl,
                            const float *scale, const float *shift, int relu, const void *res, const half_t *res_lo, float *out, half_t *out_hi,
                            half_t *out_lo, const half_t *zero_page)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = (size_t)NST * STAGE_C + 512 * sizeof(float);
    if (!attr_done) {
#define C256X_ATTR(...) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_c256_x3_kernel<__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        C256X_ATTR(0, true, false) C256X_ATTR(0, false, true) C256X_ATTR(1, true, false) C256X_ATTR(1, true, true) C256X_ATTR(2, false, true) C256X_ATTR(2, true, true)
#undef C256X_ATTR
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int ngroups = (npix + GPXC - 1) / GPXC;
    if (ngroups == 0) return;
    const int gpb = (ngroups + sfd2_slots(slots) - 1) / sfd2_slots(slots);
    const int grid = (ngroups + gpb - 1) / gpb;
#define C256X_GO(...) hipLaunchKernelGGL((conv1x1_c256_x3_kernel<__VA_ARGS__>), dim3(grid), dim3(NT1), lds, st, in, in_lo, npix, w, wl, scale, shift, relu, res, res_lo, out, out_hi, out_lo, gpb, zero_page)
    const int rmode = !res ? 0 : (res_lo ? 2 : 1);
    if (rmode == 0 && out && !out_hi) C256X_GO(0, true, false);
    else if (rmode == 0 && !out && out_hi) C256X_GO(0, false, true);
    else if (rmode == 1 && out && !out_hi) C256X_GO(1, true, false);
    else if (rmode == 1 && out && out_hi) C256X_GO(1, true, true);
    else if (rmode == 2 && !out && out_hi) C256X_GO(2, false, true);
    else if (rmode == 2 && out && out_hi) C256X_GO(2, true, true);
    else abort();
#undef C256X_GO
}

void launch_conv1x1_c256_c(hipStream_t st, const half_t *in, const half_t *in_c, int npix, const half_t *w_frag,
                           const half_t *wc_frag, const float *scale, const float *shift, int relu, const half_t *res,
                           const half_t *res_c, half_t *out, half_t *out_c, const half_t *zero_page, int sbyte, unsigned int *range, int in_r1)
// in_c == null: plain fp16 input, wc_frag = the fp16 filter residuals * 2^11; out_c == null: only the hi plane is written;
// in_r1: in_c holds residual bytes only (256 B per pixel)
{
    static bool attr_done = false;
    static int slots = 256;
    const size_t lds = in_r1 ? (size_t)SFD2_C256_R1_STAGES * STAGE_R1 + 512 * sizeof(float) : (size_t)NST * STAGE_C + 512 * sizeof(float);
    if (!attr_done) {
        const size_t lds_max = std::max((size_t)SFD2_C256_R1_STAGES * STAGE_R1, (size_t)NST * STAGE_C) + 512 * sizeof(float);
#define C256C_ATTR(R_, I_, O_) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_c256_c_kernel<R_, I_, O_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        C256C_ATTR(true, 1, true) C256C_ATTR(false, 1, true) C256C_ATTR(false, 1, false) C256C_ATTR(true, 0, true) C256C_ATTR(false, 2, false)
#undef C256C_ATTR
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int ngroups = (npix + GPXC - 1) / GPXC;
    if (ngroups == 0) return;
    const int gpb = (ngroups + sfd2_slots(slots) - 1) / sfd2_slots(slots);
    const int grid = (ngroups + gpb - 1) / gpb;
    const int sa = (sbyte & 255) * 0x01010101;
#define C256C_GO(R_, I_, O_) hipLaunchKernelGGL((conv1x1_c256_c_kernel<R_, I_, O_>), dim3(grid), dim3(NT1), lds, st, in, in_c, npix, w_frag, wc_frag, scale, shift, relu, res, res_c, out, out_c, gpb, zero_page, sa, range)
    if (in_r1) { if (in_c && !out_c && !res) C256C_GO(false, 2, false); else abort(); }
#ifdef SFD2_C256_TRACE
    if (in_r1) {
        static int dumps = 0;
        if (npix > 100000 && ++dumps == 40) {
            (void)hipStreamSynchronize(st);
            static unsigned long long hc[2][6][4];
            (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_c256_cyc), sizeof(hc));
            for (int w = 0; w < 2; ++w)
                for (int t = 0; t < 6; ++t)
                    fprintf(stderr, "c256 trace wave %d group %d: wait+barrier %lld  reads+MFMAs %lld  epilogue %lld  to next top %lld\n", w * 4, t + 4,
                            (long long)(hc[w][t][1] - hc[w][t][0]), (long long)(hc[w][t][2] - hc[w][t][1]), (long long)(hc[w][t][3] - hc[w][t][2]),
                            t < 5 ? (long long)(hc[w][t + 1][0] - hc[w][t][3]) : 0ll);
        }
    }
#endif   // ResBlock.conv1 over a residual-only input
    else if (in_c && out_c) { if (res) C256C_GO(true, 1, true); else C256C_GO(false, 1, true); }
    else if (in_c && !res) C256C_GO(false, 1, false);           // ResBlock.conv1 writing a plain t1
    else if (!in_c && out_c && res) C256C_GO(true, 0, true);   // ResBlock.conv3 reading a plain t2
    else abort();                                                   // no other combination is dispatched
#undef C256C_GO
}
