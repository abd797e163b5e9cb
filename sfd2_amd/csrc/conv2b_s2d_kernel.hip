// conv2b (3x3, stride 2, 128 -> 128 channels, nets/sfd2.py:271) of the compensated mode as a STRIDE-1 layer over conv2a's output stored
// space-to-depth.
//
// The stride-2 patch (17 x 33 input records per 8 x 16 outputs) is what forces conv3x3_rf<2,comp>'s 128-pixel tile, and with it 1.18 MB of
// filter fragments loaded per tile and CU: the layer is bound by that stream (profiles/r04_conv2b_ablations.txt).  Here conv2a writes its
// output as four parity planes [H/4][W/4][(y & 1) * 2 + (x & 1)][128] (conv3x3_pp<comp>, COMP bit 7: a different store address, nothing
// else), and an output pixel (Y, X) reads, for the original tap (ky, kx), plane (par(ky), par(kx)) at (Y + d(ky), X + d(kx)) with
// par = (1, 0, 1), d = (-1, 0, 0): a stride-1 layer whose nine taps are spread over the four planes (4 + 2 + 2 + 1).  So the tile is
// conv3x3_pp's -- 16 x 32 output pixels x 128 channels, filters through the LDS once per 512 pixels, the 18 x 34 patch of ONE plane's 32-channel
// chunk at a time -- and the K loop is a list of 32 steps per tile: (hi | corr) x 4 chunks x 4 planes, each with the plane's 1, 2 or 4 taps.
//
// Pipeline.  Per (hi | corr, chunk) group five steps: plane (1,1) with its taps of filter row 0, the same patch with those of row 2, planes
// (1,0), (0,1) (two taps each), (0,0) (one) -- 40 steps per tile, at most two taps (16 KB of filters) each.  Patches live in a ring of THREE
// buffers and are requested two patches ahead, filters in two buffers one step ahead: with two patch buffers every step waited for an HBM
// round trip that had a single step (~1.5 us of MFMAs) to hide behind -- 105 us, of which the copies alone were 80 (ablations below).  Copies
// are `buffer_load ... lds` (complete in issue order: a step waits for its own operands by COUNT and leaves the newer patch in flight; lanes
// outside the image get an offset beyond the buffer and read zeros).  One workgroup barrier per step.  Same LDS record layout / swizzle,
// fragment reads, MFMA orientation and epilogue as conv3_kernels.hip; the filter array is the layer's generic one (api_weights.hip pack_igemm:
// [hi chunks | corr chunks][tap][oc][32]).
#include "sfd2_internal.h"
#include <type_traits>
#include <stdlib.h>

#define S2_TW 32
#define S2_TH 16
#define S2_BN 128
#define S2_PW (S2_TW + 1)                       // the taps reach one record up / left only: 17 x 33 patch records
#define S2_PH (S2_TH + 1)
#define S2_NPIX (S2_PH * S2_PW)                 // 561 patch records of 64 B
#define S2_XCH ((S2_NPIX + 15) / 16)            // 36 pieces of 1 KB
#define S2_XPW 5                                // pieces per wave: waves 0 .. 3 move five, waves 4 .. 7 four
#define S2_XBYTES (S2_XCH * 1024)
#define S2_FBYTES (2 * S2_BN * 64)              // two taps x 128 filters x 64 B = 16 KB

typedef __attribute__((address_space(3))) void lds_void5_t;
typedef const __attribute__((address_space(1))) void gbl_void5_t;

// O6: the output's corr records as fp6 half-records (sfd2_epi16_fp6) for conv3a's fp6 x fp6 correction
// ABL (experiment builds, timing only): 1 = no copies inside the step loops, 2 = no fragment reads / MFMAs
// X3 (SFD2_PREC_F16X3, round 5): in / in_c are the hi / lo' PLANES of conv2a's output (conv3x3_pp<x3, planes out> with the space-to-depth store),
// wpk the filters as [4 hi chunks | 4 lo' chunks], and the step list runs THREE times over fp16 MFMAs -- hi x hi, then (the accumulators
// scaled by 2^11, exactly) hi x lo' and lo' x hi, conv3x3_pp<x3>'s arithmetic -- 60 steps and 48 patches per tile; the output leaves as hi / lo'
// planes [H4][W4][128] for conv3a.  Replaces conv3x3_rf<2, x3> on the strict mode's throughput path (192 us at 1600x1200).
template <bool O6, int ABL = 0, bool X3 = false>
__global__ __launch_bounds__(512, 2)
void conv2b_s2d_kernel(const half_t *__restrict__ in /* s2d hi plane [H4][W4][4][128] */, const half_t *__restrict__ in_c /* its corr units */,
                       int H4, int W4, const half_t *__restrict__ wpk /* [8 chunks][9][128][32] */, const float *__restrict__ scale,
                       const float *__restrict__ shift, int relu, half_t *__restrict__ out, half_t *__restrict__ out_c, int tiles_x, int n_tiles,
                       const half_t *__restrict__ zero_page, int sa, unsigned int *__restrict__ range)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                              // [3][S2_XBYTES]
    unsigned char *Fs = smem + 3 * S2_XBYTES;              // [2][S2_FBYTES]
    float *SS = reinterpret_cast<float *>(Fs + 2 * S2_FBYTES);   // scale[128], shift[128]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wch = (wave & 1) * 64;                       // 2 channel tiles
    const int wrow = (wave >> 1) * 4;                      // 4 image rows = 4 pixel tiles
    const int lrow = lane & 31, lhi = lane >> 5;
    constexpr int CIN = 512, CO = 128;
    constexpr int NPASS = X3 ? 3 : 2, NSTEP = 20 * NPASS, NPATCH = 16 * NPASS;

    for (int t = tid; t < S2_BN; t += 512) { SS[t] = scale[t]; SS[S2_BN + t] = shift[t]; }

    int a_off[2], a_sw[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int r = wch + ct * 32 + lrow;
        a_off[ct] = r * 64;
        a_sw[ct] = (r >> 2) & 3;
    }
    const int qb = wrow * S2_PW + lrow;
    // filter pieces: wave w moves rows 16 w .. 16 w + 15 of every tap's 128 x 64 B matrix
    const int frow = wave * 16 + (lane >> 2);
    const int fsrc = frow * 32 + (((lane & 3) ^ ((frow >> 2) & 3)) * 8);      // halfs within a tap's matrix

    int oy0 = 0, ox0 = 0;
    int xoff[S2_XPW];
#define S2_SETUP(tile_)                                                                                \
    {                                                                                                  \
        const int tx_ = (tile_) % tiles_x, ty_ = (tile_) / tiles_x;                                    \
        oy0 = ty_ * S2_TH; ox0 = tx_ * S2_TW;                                                          \
        _Pragma("unroll") for (int i = 0; i < S2_XPW; ++i) {                                           \
            const int piece = wave + 8 * i;      /* (i = 4: waves 0 .. 3 only, see S2_ISSUE_X) */      \
            const int q = piece * 16 + (lane >> 2);                                                    \
            const int slot = (lane & 3) ^ ((q >> 2) & 3);                                              \
            int off = (int)0x80000000;   /* beyond the buffer: zeros */                                \
            if (q < S2_NPIX) {                                                                         \
                const int py = q / S2_PW, px = q - py * S2_PW;                                         \
                const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;                                        \
                if (iy >= 0 && iy < H4 && ix >= 0 && ix < W4) off = ((iy * W4 + ix) * CIN + slot * 8) * (int)sizeof(half_t); \
            }                                                                                          \
            xoff[i] = off;                                                                             \
        }                                                                                              \
    }
    const int t_bytes = (int)((size_t)H4 * W4 * CIN * sizeof(half_t));
    const auto rs_hi = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(in), 0, t_bytes, 0x00020000);
    const auto rs_co = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(in_c), 0, t_bytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(wpk), 0, 8 * 9 * CO * 32 * (int)sizeof(half_t), 0x00020000);
    // Order within a tile: (hi | corr) x plane (1,1), (1,0), (0,1), (0,0) x chunk 0 .. 3 -- the CHUNK innermost: a pixel's 64-byte records of
    // chunks 2 j and 2 j + 1 are the two halves of one 128-byte line, and consecutive patches find the second half in the L2 the first one
    // brought it into (with the chunk outermost the halves were five steps = 6 MB of patches per XCD apart: the PMC passes counted 1.65 x
    // the tensor's bytes).
    // patch x of a tile (x = 0 .. 31): corr = x >> 4, memory plane 3 - ((x >> 2) & 3), chunk x & 3; five (waves 4 .. 7: four) copies per wave
    // (X3: x = 0 .. 47, pass = x >> 4: the hi plane for passes 0 and 1, the lo' plane for pass 2)
#define S2_ISSUE_X(x_, xb_)                                                                            \
    {                                                                                                  \
        const int g_ = (X3 ? ((x_) >> 5) : ((x_) >> 4)) << 2, pl_ = 3 - (((x_) >> 2) & 3);     /* (g_ >> 2 = the second input plane) */ \
        const int so_ = (pl_ * 128 + ((x_) & 3) * 32) * (int)sizeof(half_t);                            \
        _Pragma("unroll") for (int i = 0; i < S2_XPW; ++i) {                                           \
            const int pc_ = wave + 8 * i;                                                              \
            if (pc_ >= S2_XCH) continue;         /* wave-uniform */                                    \
            if ((g_ >> 2) != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_co, (lds_void5_t *)(Xs + (xb_)*S2_XBYTES + pc_ * 1024), 16, xoff[i], so_, 0, 0); \
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_hi, (lds_void5_t *)(Xs + (xb_)*S2_XBYTES + pc_ * 1024), 16, xoff[i], so_, 0, 0); \
        }                                                                                              \
    }
    // step f (f = 0 .. 39): corr = f / 20, r = f % 20; r < 8: plane (1,1), chunk r >> 1, e = r & 1 (filter row 0 / 2); else plane index
    // (r - 8) >> 2, chunk (r - 8) & 3, e = 2 + plane index.  e -> original taps (0, 2), (6, 8), (1, 7), (3, 5), (4); ONE copy per wave and tap
#define S2_STEP_MAP(f_, cr_, e_, k_)                                                                   \
        const int cr_ = (f_) >= 40 ? 2 : (f_) >= 20 ? 1 : 0, r__##e_ = (f_) - cr_ * 20;   /* the pass */     \
        const int e_ = r__##e_ < 8 ? (r__##e_ & 1) : 2 + ((r__##e_ - 8) >> 2);                          \
        const int k_ = r__##e_ < 8 ? (r__##e_ >> 1) : ((r__##e_ - 8) & 3);
#define S2_ISSUE_F(f_, fb_)                                                                            \
    {                                                                                                  \
        S2_STEP_MAP(f_, c_, e_, kk_)                                                                   \
        const int g_ = (X3 ? (c_ == 1 ? 4 : 0) : c_ * 4) + kk_;   /* X3: hi, lo', hi filter chunks */     \
        const int t0_ = e_ == 0 ? 0 : e_ == 1 ? 6 : e_ == 2 ? 1 : e_ == 3 ? 3 : 4;                      \
        const int t1_ = e_ == 0 ? 2 : e_ == 1 ? 8 : e_ == 2 ? 7 : 5;                                   \
        const int sb_ = g_ * 9 * CO * 32 * (int)sizeof(half_t);                                        \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void5_t *)(Fs + (fb_)*S2_FBYTES + wave * 1024), 16, fsrc * 2, sb_ + t0_ * CO * 32 * 2, 0, 0); \
        if (e_ != 4)                                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void5_t *)(Fs + (fb_)*S2_FBYTES + S2_BN * 64 + wave * 1024), 16, fsrc * 2, sb_ + t1_ * CO * 32 * 2, 0, 0); \
    }

    int tile = blockIdx.x;
    S2_SETUP(tile)
    S2_ISSUE_X(0, 0)
    S2_ISSUE_X(1, 1)
    S2_ISSUE_F(0, 0)
    unsigned int smax = 0;
    int xb = 0;                                            // ring buffer of the patch at hand (patch x + 1 in xb + 1, x + 2 goes to xb + 2)

    for (;;) {
        f32x16_t acc[2][4];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < n_tiles;
        const int eoy0 = oy0, eox0 = ox0;
        bool lax = false;                                  // the previous step requested a patch BEHIND this step's filters: those copies (five or four per wave) may stay in flight

        // Start of step f: this step's filters (requested a step ago, in front of that step's patch request) and patch (two patches ago) have
        // landed -- every wave waits for its own copies, then the block; every wave is also done reading what the copies issued below
        // overwrite (the other filter buffer: step f - 1; the third patch buffer: the previous patch).  Then the requests: filters of step
        // f + 1 FIRST, then (first step of a patch) the patch after next.
#define S2_STEP_PROLOGUE(f_)                                                                           \
            if (lax && wave < 4) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
            else if (lax) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");     \
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");              \
            S2_STEP_MAP(f_, cr, e, kc)                                                                  \
            const int fbuf = (f_) & 1;                                                                  \
            lax = false;                                                                               \
            if (!(ABL & 1)) {                                                                          \
                if ((f_) + 1 < NSTEP) { S2_ISSUE_F((f_) + 1, fbuf ^ 1) }                                \
                else if (has_next) { S2_ISSUE_F(0, fbuf ^ 1) }                                          \
                if (e != 1) {   /* first step of patch x = 16 corr + 4 plane index + chunk */           \
                    const int x2 = 16 * cr + (e >= 2 ? 4 * (e - 1) : 0) + kc + 2;                       \
                    const int b2 = xb >= 1 ? xb - 1 : 2;   /* (xb + 2) % 3 */                           \
                    if (x2 < NPATCH) { S2_ISSUE_X(x2, b2) lax = true; }                                 \
                    else if (has_next) {                                                               \
                        if (x2 == NPATCH) { S2_SETUP(next) }                                           \
                        S2_ISSUE_X(x2 - NPATCH, b2)                                                    \
                        lax = true;                                                                    \
                    }                                                                                  \
                }                                                                                      \
            }                                                                                          \
            const int ntap = e == 4 ? 1 : 2;                                                            \
            /* tap j of the step: patch shift (rows, columns) */                                        \
            const int ro0 = (e == 0 || e == 2) ? 0 : 1, co0 = (e == 2 || e == 4) ? 1 : 0;               \
            const int ro1 = e == 0 ? 0 : 1, co1 = 1;                                                    \
            const unsigned char *xs = Xs + xb * S2_XBYTES;                                             \
            const unsigned char *fs = Fs + fbuf * S2_FBYTES;
#define S2_STEP_EPILOGUE()                                                                             \
            if (e != 0) xb = xb == 2 ? 0 : xb + 1;         /* (step 0 of a group shares its patch with step 1) */
#define S2_FP16_TAPS() \
_Pragma("unroll 1") \
            for (int j = 0; j < ((ABL & 2) ? 0 : ntap); ++j) { \
                const unsigned char *ft = fs + j * (S2_BN * 64); \
                const int qv = qb + (j ? ro1 : ro0) * S2_PW + (j ? co1 : co0); \
                h8_t fa[2][2], fb[2][4]; \
_Pragma("unroll") \
                for (int ct = 0; ct < 2; ++ct) \
_Pragma("unroll") \
                    for (int kk = 0; kk < 2; ++kk) \
                        fa[kk][ct] = *reinterpret_cast<const h8_t *>(ft + a_off[ct] + (((kk * 2 + lhi) ^ a_sw[ct]) << 4)); \
_Pragma("unroll") \
                for (int pr = 0; pr < 4; ++pr) { \
                    const int q = qv + pr * S2_PW; \
                    const int sw = (q >> 2) & 3; \
_Pragma("unroll") \
                    for (int kk = 0; kk < 2; ++kk) \
                        fb[kk][pr] = *reinterpret_cast<const h8_t *>(xs + q * 64 + (((kk * 2 + lhi) ^ sw) << 4)); \
                } \
_Pragma("unroll") \
                for (int kk = 0; kk < 2; ++kk) \
_Pragma("unroll") \
                    for (int ct = 0; ct < 2; ++ct) \
_Pragma("unroll") \
                        for (int pr = 0; pr < 4; ++pr) \
                            acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][ct], fb[kk][pr], acc[ct][pr], 0, 0, 0); \
            } \

#pragma unroll 1
        for (int f = 0; f < 20; ++f) {
            S2_STEP_PROLOGUE(f)
            S2_FP16_TAPS()
            S2_STEP_EPILOGUE()
        }
        if constexpr (X3) {
            // hi x hi is done: scale the sums by 2^11 (exact) so that the cross terms -- whose lo' operands carry that factor -- join the
            // same accumulators; 2^-11 goes into the epilogue
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] *= 2048.0f;
#pragma unroll 1
            for (int f = 20; f < 60; ++f) {
                S2_STEP_PROLOGUE(f)
                S2_FP16_TAPS()
                S2_STEP_EPILOGUE()
            }
        } else {
#pragma unroll 1
        for (int f = 20; f < 40; ++f) {
            S2_STEP_PROLOGUE(f)
#pragma unroll 1
            for (int j = 0; j < ((ABL & 2) ? 0 : ntap); ++j) {
                const unsigned char *ft = fs + j * (S2_BN * 64);
                const int qv = qb + (j ? ro1 : ro0) * S2_PW + (j ? co1 : co0);
                v8i_t fac[2], frc[4];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    fac[ct] = sfd2_cat8(*reinterpret_cast<const h8_t *>(ft + a_off[ct] + (((0 * 2 + lhi) ^ a_sw[ct]) << 4)),
                                        *reinterpret_cast<const h8_t *>(ft + a_off[ct] + (((1 * 2 + lhi) ^ a_sw[ct]) << 4)));
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const int q = qv + pr * S2_PW;
                    const int sw = (q >> 2) & 3;
                    frc[pr] = sfd2_cat8(*reinterpret_cast<const h8_t *>(xs + q * 64 + (((0 * 2 + lhi) ^ sw) << 4)),
                                        *reinterpret_cast<const h8_t *>(xs + q * 64 + (((1 * 2 + lhi) ^ sw) << 4)));
                }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr)
                        acc[ct][pr] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fac[ct], frc[pr], acc[ct][pr], 0, 0, 0, sa, 0, 0x7f7f7f7f);
                // (the scaled MFMA's destination is untied: pin the accumulators so that none is copied at the loop edge)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) asm volatile("" : "+v"(acc[ct][pr]));
            }
            S2_STEP_EPILOGUE()
        }
        }
#undef S2_STEP_PROLOGUE
#undef S2_STEP_EPILOGUE
#undef S2_FP16_TAPS
        // (the epilogue's stores are younger than every copy in flight: the next tile's first wait is a full one)

        // epilogue (conv3_kernels.hip): y = acc * scale + shift, ReLU, hi plane + corr records, 16-byte stores
        float mx = 0.0f;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int oy = eoy0 + wrow + pr, ox = eox0 + lrow;
            const bool inb = oy < H4 && ox < W4;
            const size_t pix = (size_t)(inb ? oy : 0) * W4 + (inb ? ox : 0);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int cl = wch + ct * 32 + 4 * lhi;
                const size_t ob = pix * CO + wch + ct * 32;
                if constexpr (X3) {      // hi / lo' planes (x3_split's arithmetic, as conv3x3_pp<x3, planes out>)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        uint2 pk[2], ck[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int q = 2 * m + j;
                            const float4 sc = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                            const float4 sh = *reinterpret_cast<const float4 *>(SS + S2_BN + cl + 8 * q);
                            constexpr float k = 1.0f / 2048.0f;
                            const float lo = relu ? 0.0f : -__builtin_huge_valf();
                            float v0 = acc[ct][pr][4 * q + 0] * k * sc.x + sh.x, v1 = acc[ct][pr][4 * q + 1] * k * sc.y + sh.y;
                            float v2 = acc[ct][pr][4 * q + 2] * k * sc.z + sh.z, v3 = acc[ct][pr][4 * q + 3] * k * sc.w + sh.w;
                            v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
                            const h4_t hv = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                            const h4_t lv = {(half_t)((v0 - (float)hv[0]) * 2048.0f), (half_t)((v1 - (float)hv[1]) * 2048.0f),
                                             (half_t)((v2 - (float)hv[2]) * 2048.0f), (half_t)((v3 - (float)hv[3]) * 2048.0f)};
                            __builtin_memcpy(&pk[j], &hv, 8);
                            __builtin_memcpy(&ck[j], &lv, 8);
                        }
                        const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                        const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                        if (inb) {
                            *reinterpret_cast<uint4 *>(out + ob + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                            *reinterpret_cast<uint4 *>(out_c + ob + 8 * (2 * m + lhi)) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                        }
                    }
                } else if constexpr (O6) {
                    float4 sc4[4], sh4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sc4[q] = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                        sh4[q] = *reinterpret_cast<const float4 *>(SS + S2_BN + cl + 8 * q);
                    }
                    uint2 hv4[4];
                    uint4 r0, r1;
                    sfd2_epi16_fp6(acc[ct][pr], sc4, sh4, relu ? 0.0f : -SFD2_C_SAT, hv4, r0, r1, mx, inb);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const auto t0 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].x, hv4[2 * m + 1].x, false, false);
                        const auto t1 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].y, hv4[2 * m + 1].y, false, false);
                        if (inb) *reinterpret_cast<uint4 *>(out + ob + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                    }
                    if (inb) {
                        *reinterpret_cast<uint4 *>(out_c + ob + 8 * lhi) = r0;
                        *reinterpret_cast<uint4 *>(out_c + ob + 16 + 8 * lhi) = r1;
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        uint2 pk[2], ck[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int q = 2 * m + j;
                            const float4 sc = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                            const float4 sh = *reinterpret_cast<const float4 *>(SS + S2_BN + cl + 8 * q);
                            sfd2_epi4<false>(acc[ct][pr][4 * q + 0], acc[ct][pr][4 * q + 1], acc[ct][pr][4 * q + 2], acc[ct][pr][4 * q + 3], sc, sh,
                                             sc, relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j], mx, inb);
                        }
                        const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                        const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                        const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                        const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                        if (inb) {
                            *reinterpret_cast<uint4 *>(out + ob + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                            *reinterpret_cast<uint4 *>(out_c + ob + 8 * (2 * m + lhi)) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                        }
                    }
                }
            }
        }
        if (!X3) { const unsigned int wb = sfd2_wave_max_bits(mx); smax = wb > smax ? wb : smax; }
        if (!has_next) break;
        tile = next;
    }
    if (!X3 && range != nullptr) sfd2_range_commit(range, smax);
#undef S2_SETUP
#undef S2_ISSUE_X
#undef S2_ISSUE_F
#undef S2_STEP_MAP
}

// does the s2d form serve this geometry?  (conv2a's output H2 x W2 must split into whole 2 x 2 cells)
bool conv2b_s2d_serves(int H2, int W2, int Cin, int CoutP)
{
    return (H2 % 2) == 0 && (W2 % 2) == 0 && Cin == 128 && CoutP == 128 && H2 >= 2 && W2 >= 2 &&
           (long long)H2 * W2 * 128 * (long long)sizeof(half_t) < (1ll << 31);      // (buffer offsets are 32-bit)
}

void launch_conv2b_s2d(hipStream_t st, const half_t *in, const half_t *in_c, int H4, int W4, const half_t *wpk, const float *scale,
                       const float *shift, int relu, half_t *out, half_t *out_c, const half_t *zero_page, int sbyte, unsigned int *range, int fmt6)
{
    constexpr size_t lds = (size_t)3 * S2_XBYTES + (size_t)2 * S2_FBYTES + 2 * S2_BN * sizeof(float);
    static bool attr_done = false;
    static int slots = 256;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2b_s2d_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2b_s2d_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int tiles_x = (W4 + S2_TW - 1) / S2_TW, tiles_y = (H4 + S2_TH - 1) / S2_TH;
    const int n_tiles = tiles_x * tiles_y;
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    const int sa = (sbyte & 255) * 0x01010101;
#ifdef SFD2_EXPERIMENTS
    if (const char *ab = sfd2_env("SFD2_S2D_ABL")) {
        const int a = atoi(ab);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2b_s2d_kernel<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2b_s2d_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv2b_s2d_kernel<true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (a == 1) hipLaunchKernelGGL((conv2b_s2d_kernel<true, 1>), dim3(grid), dim3(512), lds, st, in, in_c, H4, W4, wpk, scale, shift, relu, out, out_c, tiles_x, n_tiles, zero_page, sa, range);
        if (a == 2) hipLaunchKernelGGL((conv2b_s2d_kernel<true, 2>), dim3(grid), dim3(512), lds, st, in, in_c, H4, W4, wpk, scale, shift, relu, out, out_c, tiles_x, n_tiles, zero_page, sa, range);
        if (a == 3) hipLaunchKernelGGL((conv2b_s2d_kernel<true, 3>), dim3(grid), dim3(512), lds, st, in, in_c, H4, W4, wpk, scale, shift, relu, out, out_c, tiles_x, n_tiles, zero_page, sa, range);
        if (a >= 1 && a <= 3) return;
    }
#endif
    if (fmt6 & 2)
        hipLaunchKernelGGL(conv2b_s2d_kernel<true>, dim3(grid), dim3(512), lds, st, in, in_c, H4, W4, wpk, scale, shift, relu, out, out_c, tiles_x,
                           n_tiles, zero_page, sa, range);
    else
        hipLaunchKernelGGL(conv2b_s2d_kernel<false>, dim3(grid), dim3(512), lds, st, in, in_c, H4, W4, wpk, scale, shift, relu, out, out_c, tiles_x,
                           n_tiles, zero_page, sa, range);
}

// SFD2_PREC_F16X3: conv2b on hi / lo' planes stored space-to-depth (conv3x3_pp<x3, planes out, s2d>), planes out.  wpk = the layer's f16x3
// filter planes, [4 hi chunks | 4 lo' chunks][9][128][32] (api_network.hip convf: x3_split_planes of the packed fp32 filters).
void launch_conv2b_s2d_x3(hipStream_t st, const half_t *in_hi, const half_t *in_lo, int H4, int W4, const half_t *wpk, const float *scale,
                          const float *shift, int relu, half_t *out_hi, half_t *out_lo, const half_t *zero_page)
{
    constexpr size_t lds = (size_t)3 * S2_XBYTES + (size_t)2 * S2_FBYTES + 2 * S2_BN * sizeof(float);
    static bool attr_done = false;
    static int slots = 256;
    auto kern = conv2b_s2d_kernel<false, 0, true>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int tiles_x = (W4 + S2_TW - 1) / S2_TW, tiles_y = (H4 + S2_TH - 1) / S2_TH;
    const int n_tiles = tiles_x * tiles_y;
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, in_hi, in_lo, H4, W4, wpk, scale, shift, relu, out_hi, out_lo, tiles_x, n_tiles,
                       zero_page, 0, nullptr);
}
