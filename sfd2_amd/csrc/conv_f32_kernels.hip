// Strict (fp32) conv stack: the same ResSegNetV2 layers as conv_kernels.hip, computed in exact fp32
// on the f32-input MFMA (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 FMA chain, 157 TFLOP/s peak)
// with fp32 activations in HBM.  This is the parity mode: it differs from the fp32 reference only
// by summation order (~1e-6), so key points, stability classes and descriptors reproduce the
// reference's to fp32 round-off; the fp16 kernels are the throughput mode.
//
// Replaces the same reference modules: nets/sfd2.py:259-303 as executed by det (:313-347).
#include "sfd2_internal.h"
#include <stdlib.h>

#define TW 32
#define TH 4
#define CC 32
#define PIXF 36   // floats per pixel record in LDS: 32 + 4 pad (144 B: conflict-free ds_read_b128)
#define NT 256

__device__ __forceinline__ int xcd_swizzle_f(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM conv, fp32.  Block = 4 waves, tile 4 x 32 px x BN channels (BN = 64 or 128; a
// 256-channel layer is two channel tiles).  Filter tiles double-buffered through registers, the
// input patch single-buffered (re-staged once per 32-channel chunk).
// A lane's 16-byte fragment read holds 4 consecutive k of an 8-wide k block (k = 4*(lane>>5) + j);
// MFMA j of the block consumes element j of both operands, so A and B always pair the same k.
template <int KS, int STRIDE, int BN, bool HAS_RES>
__global__ __launch_bounds__(NT)
void conv_igemm_f32_kernel(const float *__restrict__ in, int H, int W, int Cin,
                           const float *__restrict__ wpk, const float *__restrict__ scale,
                           const float *__restrict__ shift, int CoutP, int relu,
                           const float *__restrict__ res, float *__restrict__ out,
                           int Ho, int Wo, int tiles_x)
{
    constexpr int T = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int PH = (TH - 1) * STRIDE + KS;
    constexpr int PW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = PH * PW;
    constexpr int XPIECES = NPIX * 8;                  // 16-byte pieces of one patch chunk (32 floats / pixel)
    constexpr int WPIECES = BN * 8;
    constexpr int WP = WPIECES / NT;                   // BN 64 -> 2, 128 -> 4
    constexpr int WAVES_CH = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_PX = 4 / WAVES_CH;
    constexpr int CH_T = BN / WAVES_CH / 32;
    constexpr int PX_T = TH / WAVES_PX;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XB = (KS == 1) ? 2 : 1;              // 1x1: the patch is double-buffered and prefetched like the filters
    float *Xs = reinterpret_cast<float *>(smem);       // [XB][NPIX][PIXF]
    float *Ws = Xs + XB * NPIX * PIXF;                 // [2][BN][PIXF]
    float *SS = Ws + 2 * BN * PIXF;                    // scale[BN], shift[BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wch = (wave % WAVES_CH) * (CH_T * 32);
    const int wrow = (wave / WAVES_CH) * PX_T;
    const int n_tiles_n = CoutP / BN;
    const int swz = xcd_swizzle_f(blockIdx.x, gridDim.x);
    const int tn = swz % n_tiles_n;
    const int tsp = swz / n_tiles_n;
    const int tx = tsp % tiles_x, ty = tsp / tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;

    float4 wr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);

#define STAGE_X(chunk_)                                                                                \
    for (int p = tid; p < XPIECES; p += NT) {                                                          \
        const int q = p >> 3, part = p & 7;                                                            \
        const int py = q / PW, px = q - py * PW;                                                       \
        const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                          \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)                                                    \
            v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * Cin + (chunk_)*CC + part * 4)); \
        *reinterpret_cast<float4 *>(Xs + q * PIXF + part * 4) = v;                                     \
    }
    // 1x1 layers: every step is a new chunk; the next chunk's pieces are fetched into registers before the MFMAs and stored
    // after them into the other buffer (one barrier per step, no exposed global-load latency; see conv_igemm_x3_kernel)
    constexpr int XP = (XPIECES + NT - 1) / NT;
    float4 xr[XP];
#define LOAD_X(chunk_)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                   \
        const int p = tid + i * NT, q = p >> 3, part = p & 7;                                          \
        const int py = q / PW, px = q - py * PW;                                                       \
        const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                          \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        if (p < XPIECES && iy >= 0 && iy < H && ix >= 0 && ix < W)                                     \
            v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * Cin + (chunk_)*CC + part * 4)); \
        xr[i] = v;                                                                                     \
    }
#define STORE_X(buf_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                   \
        const int p = tid + i * NT, q = p >> 3, part = p & 7;                                          \
        if (p < XPIECES) *reinterpret_cast<float4 *>(Xs + ((buf_)*NPIX + q) * PIXF + part * 4) = xr[i]; \
    }
#define LOAD_W(step_)                                                                                  \
    {                                                                                                  \
        const float *wbase_ = wpk + ((size_t)(step_)*CoutP + n0) * CC;                                 \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                                 \
            wr[i] = *reinterpret_cast<const float4 *>(wbase_ + (size_t)(tid + i * NT) * 4);            \
    }
#define STORE_W(buf_)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                   \
        const int p = tid + i * NT, row = p >> 3, part = p & 7;                                        \
        *reinterpret_cast<float4 *>(Ws + ((buf_)*BN + row) * PIXF + part * 4) = wr[i];                 \
    }

    f32x16_t acc[CH_T][PX_T];
#pragma unroll
    for (int a = 0; a < CH_T; ++a)
#pragma unroll
        for (int b = 0; b < PX_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int NS = (Cin / CC) * T;
    STAGE_X(0)
    LOAD_W(0)
    STORE_W(0)
    if (tid < BN) { SS[tid] = scale[n0 + tid]; SS[BN + tid] = shift[n0 + tid]; }
    __syncthreads();

    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    int chunk = 0, tap = 0;
    for (int s = 0; s < NS; ++s) {
        const int wb = s & 1;
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == T) { ntap = 0; ++nchunk; }
        const bool has_next = (s + 1 < NS);
        const bool new_chunk = has_next && (ntap == 0);
        if (has_next) LOAD_W(s + 1)
        if ((KS == 1 && has_next) || new_chunk) { LOAD_X(nchunk) }   // 1x1: every step; 3x3: at a chunk's last tap

        const int ky = tap / KS, kx = tap - ky * KS;
        const float *ws = Ws + wb * BN * PIXF;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 a[CH_T], b[PX_T];
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
                a[ct] = *reinterpret_cast<const float4 *>(ws + (wch + ct * 32 + lrow) * PIXF + kb * 8 + lk);
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr) {
                const int q = ((wrow + pr) * STRIDE + ky) * PW + lrow * STRIDE + kx;
                b[pr] = *reinterpret_cast<const float4 *>(Xs + ((KS == 1 ? (s & 1) * NPIX : 0) + q) * PIXF + kb * 8 + lk);
            }
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int pr = 0; pr < PX_T; ++pr) {
                    acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct].x, b[pr].x, acc[ct][pr], 0, 0, 0);
                    acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct].y, b[pr].y, acc[ct][pr], 0, 0, 0);
                    acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct].z, b[pr].z, acc[ct][pr], 0, 0, 0);
                    acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct].w, b[pr].w, acc[ct][pr], 0, 0, 0);
                }
        }
        if (has_next) { STORE_W(wb ^ 1) }
        if (KS == 1 && has_next) { STORE_X((s + 1) & 1) }
        __syncthreads();
        if (KS != 1 && new_chunk) {          // every wave is past its reads of the (single) patch buffer: store the prefetched chunk
            STORE_X(0)
            __syncthreads();
        }
        tap = ntap;
        chunk = nchunk;
    }
#undef STAGE_X
#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W

#pragma unroll
    for (int pr = 0; pr < PX_T; ++pr) {
        const int oy = oy0 + wrow + pr, ox = ox0 + lrow;
        if (oy < Ho && ox < Wo) {
            const size_t pix = (size_t)oy * Wo + ox;
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wch + ct * 32 + 8 * q + 4 * (lane >> 5);
                    const float4 sc = *reinterpret_cast<const float4 *>(SS + cl);
                    const float4 sh = *reinterpret_cast<const float4 *>(SS + BN + cl);
                    float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                    float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                    float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                    float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                    const size_t o = pix * CoutP + n0 + cl;
                    if (HAS_RES) {
                        const float4 r = *reinterpret_cast<const float4 *>(res + o);
                        v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                    }
                    if (relu) {
                        v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                    }
                    *reinterpret_cast<float4 *>(out + o) = make_float4(v0, v1, v2, v3);
                }
        }
    }
}

template <int KS, int STRIDE, int BN, bool HAS_RES>
static void launch_f32_t(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk, const float *scale,
                         const float *shift, int CoutP, int relu, const float *res, float *out, int Ho, int Wo)
{
    constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    constexpr size_t lds = (size_t)((KS == 1 ? 2 : 1) * PH * PW + 2 * BN) * PIXF * sizeof(float) + (size_t)2 * BN * sizeof(float);
    static bool attr_done = false;
    auto kern = conv_igemm_f32_kernel<KS, STRIDE, BN, HAS_RES>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    hipLaunchKernelGGL(kern, dim3(tiles_x * tiles_y * (CoutP / BN)), dim3(NT), lds, st, in, H, W, Cin, wpk, scale, shift,
                       CoutP, relu, res, out, Ho, Wo, tiles_x);
}

void launch_conv_igemm_f32(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk,
                           const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                           const float *residual, float *out, int Ho, int Wo)
{
#define SFD2_F32(KS_, ST_, BN_)                                                                                        \
    do {                                                                                                              \
        if (residual) launch_f32_t<KS_, ST_, BN_, true>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo); \
        else launch_f32_t<KS_, ST_, BN_, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo);         \
    } while (0)
    const bool b128 = (CoutP % 128 == 0);
    if (ks == 3 && stride == 1) { if (b128) SFD2_F32(3, 1, 128); else SFD2_F32(3, 1, 64); }
    else if (ks == 3 && stride == 2) { if (b128) SFD2_F32(3, 2, 128); else SFD2_F32(3, 2, 64); }
    else if (ks == 1 && stride == 1) { if (b128) SFD2_F32(1, 1, 128); else SFD2_F32(1, 1, 64); }
    else abort();
#undef SFD2_F32
}

// ---------------------------------------------------------------------------------------------
// The same implicit GEMM on the fp16 matrix path, three passes ("f16x3", precision mode SFD2_PREC_F16X3): every fp32
// operand x is split while it is staged into LDS into hi = fp16(x) and lo = fp16((x - hi) * 2048), and
//     x * w  ~=  hi_x * hi_w  +  (hi_x * lo_w + lo_x * hi_w) / 2048          (the lo * lo term, <= 2^-22 relative, is dropped)
// with both sums accumulated in fp32 by v_mfma_f32_32x32x16_f16.  Activations and filters stay fp32 in HBM: nothing but
// this kernel changes against the strict mode.  The factor 2048 keeps lo in fp16's normal range whenever x is (so it does
// not matter whether the matrix pipe flushes fp16 subnormals).  Per product the error is ~2^-22 relative against fp32's
// 2^-24; accumulation is fp32 as before.  Cost: 3 MFMAs of 32 cycles per 16-wide k slice instead of 8 of 64.
#define PIXH 40   // halves per pixel record in LDS: 32 + 8 pad (80 B: conflict-free ds_read_b128)
#define X3_SCALE 2048.0f

__device__ __forceinline__ void x3_split(const float4 v, h4_t &hi, h4_t &lo)
{
    hi[0] = (half_t)v.x; hi[1] = (half_t)v.y; hi[2] = (half_t)v.z; hi[3] = (half_t)v.w;
    lo[0] = (half_t)((v.x - (float)hi[0]) * X3_SCALE);
    lo[1] = (half_t)((v.y - (float)hi[1]) * X3_SCALE);
    lo[2] = (half_t)((v.z - (float)hi[2]) * X3_SCALE);
    lo[3] = (half_t)((v.w - (float)hi[3]) * X3_SCALE);
}

template <int KS, int STRIDE, int BN, bool HAS_RES>
__global__ __launch_bounds__(NT)
void conv_igemm_x3_kernel(const float *__restrict__ in, int H, int W, int Cin,
                          const float *__restrict__ wpk, const float *__restrict__ scale,
                          const float *__restrict__ shift, int CoutP, int relu,
                          const float *__restrict__ res, float *__restrict__ out,
                          int Ho, int Wo, int tiles_x)
{
    constexpr int T = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int PH = (TH - 1) * STRIDE + KS;
    constexpr int PW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = PH * PW;
    constexpr int XPIECES = NPIX * 8;                  // 4-channel pieces of one patch chunk (32 channels / pixel)
    constexpr int WPIECES = BN * 8;
    constexpr int WP = WPIECES / NT;                   // BN 64 -> 2, 128 -> 4
    constexpr int WAVES_CH = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_PX = 4 / WAVES_CH;
    constexpr int CH_T = BN / WAVES_CH / 32;
    constexpr int PX_T = TH / WAVES_PX;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XB = (KS == 1) ? 2 : 1;              // 1x1: the patch is double-buffered and prefetched like the filters
    half_t *Xh = reinterpret_cast<half_t *>(smem);     // [XB][NPIX][PIXH]
    half_t *Xl = Xh + XB * NPIX * PIXH;
    half_t *Wh = Xl + XB * NPIX * PIXH;                // [2][BN][PIXH]
    half_t *Wl = Wh + 2 * BN * PIXH;
    float *SS = reinterpret_cast<float *>(Wl + 2 * BN * PIXH);   // scale[BN], shift[BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wch = (wave % WAVES_CH) * (CH_T * 32);
    const int wrow = (wave / WAVES_CH) * PX_T;
    const int n_tiles_n = CoutP / BN;
    const int swz = xcd_swizzle_f(blockIdx.x, gridDim.x);
    const int tn = swz % n_tiles_n;
    const int tsp = swz / n_tiles_n;
    const int tx = tsp % tiles_x, ty = tsp / tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;

    float4 wr[WP];
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = make_float4(0.f, 0.f, 0.f, 0.f);

#define X3_STAGE_X(chunk_)                                                                             \
    for (int p = tid; p < XPIECES; p += NT) {                                                          \
        const int q = p >> 3, part = p & 7;                                                            \
        const int py = q / PW, px = q - py * PW;                                                       \
        const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                          \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)                                                    \
            v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * Cin + (chunk_)*CC + part * 4)); \
        h4_t hi, lo;                                                                                   \
        x3_split(v, hi, lo);                                                                           \
        *reinterpret_cast<h4_t *>(Xh + q * PIXH + part * 4) = hi;                                      \
        *reinterpret_cast<h4_t *>(Xl + q * PIXH + part * 4) = lo;                                      \
    }
    // 1x1 layers: every step is a new 32-channel chunk.  Re-staging the patch between two barriers after the MFMAs (as the
    // 3x3 layers do once per nine taps) exposes a global-load latency per step; here the next chunk's pieces are fetched
    // into registers before the MFMAs and split / stored after them, into the other buffer: one barrier per step.
    constexpr int XP = (XPIECES + NT - 1) / NT;
    float4 xr[XP];
#define X3_LOAD_X(chunk_)                                                                              \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                   \
        const int p = tid + i * NT, q = p >> 3, part = p & 7;                                          \
        const int py = q / PW, px = q - py * PW;                                                       \
        const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                          \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                    \
        if (p < XPIECES && iy >= 0 && iy < H && ix >= 0 && ix < W)                                     \
            v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * Cin + (chunk_)*CC + part * 4)); \
        xr[i] = v;                                                                                     \
    }
#define X3_STORE_X(buf_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                   \
        const int p = tid + i * NT, q = p >> 3, part = p & 7;                                          \
        if (p < XPIECES) {                                                                             \
            h4_t hi, lo;                                                                               \
            x3_split(xr[i], hi, lo);                                                                   \
            *reinterpret_cast<h4_t *>(Xh + ((buf_)*NPIX + q) * PIXH + part * 4) = hi;                  \
            *reinterpret_cast<h4_t *>(Xl + ((buf_)*NPIX + q) * PIXH + part * 4) = lo;                  \
        }                                                                                              \
    }
#define X3_LOAD_W(step_)                                                                               \
    {                                                                                                  \
        const float *wbase_ = wpk + ((size_t)(step_)*CoutP + n0) * CC;                                 \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                                 \
            wr[i] = *reinterpret_cast<const float4 *>(wbase_ + (size_t)(tid + i * NT) * 4);            \
    }
#define X3_STORE_W(buf_)                                                                               \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                   \
        const int p = tid + i * NT, row = p >> 3, part = p & 7;                                        \
        /* the filters arrive pre-split (x3_split_kernel): 16 bytes = 4 hi + 4 lo halves of one float4 */ \
        *reinterpret_cast<float2 *>(Wh + ((buf_)*BN + row) * PIXH + part * 4) = make_float2(wr[i].x, wr[i].y); \
        *reinterpret_cast<float2 *>(Wl + ((buf_)*BN + row) * PIXH + part * 4) = make_float2(wr[i].z, wr[i].w); \
    }

    f32x16_t accm[CH_T][PX_T], accl[CH_T][PX_T];       // hi * hi  |  (hi * lo + lo * hi) * 2048
#pragma unroll
    for (int a = 0; a < CH_T; ++a)
#pragma unroll
        for (int b = 0; b < PX_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accm[a][b][r] = 0.0f; accl[a][b][r] = 0.0f; }

    const int NS = (Cin / CC) * T;
    X3_STAGE_X(0)
    X3_LOAD_W(0)
    X3_STORE_W(0)
    if (tid < BN) { SS[tid] = scale[n0 + tid]; SS[BN + tid] = shift[n0 + tid]; }
    __syncthreads();

    const int lrow = lane & 31, lk = (lane >> 5) * 8;
    int chunk = 0, tap = 0;
    for (int s = 0; s < NS; ++s) {
        const int wb = s & 1;
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == T) { ntap = 0; ++nchunk; }
        const bool has_next = (s + 1 < NS);
        const bool new_chunk = has_next && (ntap == 0);
        if (has_next) X3_LOAD_W(s + 1)
        // next patch chunk -> registers, ahead of this step's MFMAs (1x1: every step; 3x3: at a chunk's last tap)
        if ((KS == 1 && has_next) || new_chunk) { X3_LOAD_X(nchunk) }

        const int ky = tap / KS, kx = tap - ky * KS;
        const half_t *wh = Wh + wb * BN * PIXH, *wl = Wl + wb * BN * PIXH;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            h8_t ah[CH_T], al[CH_T], bh[PX_T], bl[PX_T];
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct) {
                const int o = (wch + ct * 32 + lrow) * PIXH + kb * 16 + lk;
                ah[ct] = *reinterpret_cast<const h8_t *>(wh + o);
                al[ct] = *reinterpret_cast<const h8_t *>(wl + o);
            }
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr) {
                const int q = ((wrow + pr) * STRIDE + ky) * PW + lrow * STRIDE + kx;
                const int xq = (KS == 1 ? (s & 1) * NPIX : 0) + q;
                bh[pr] = *reinterpret_cast<const h8_t *>(Xh + xq * PIXH + kb * 16 + lk);
                bl[pr] = *reinterpret_cast<const h8_t *>(Xl + xq * PIXH + kb * 16 + lk);
            }
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int pr = 0; pr < PX_T; ++pr) {
                    accm[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bh[pr], accm[ct][pr], 0, 0, 0);
                    accl[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct], bl[pr], accl[ct][pr], 0, 0, 0);
                    accl[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ct], bh[pr], accl[ct][pr], 0, 0, 0);
                }
        }
        if (has_next) { X3_STORE_W(wb ^ 1) }
        if (KS == 1 && has_next) { X3_STORE_X((s + 1) & 1) }
        __syncthreads();
        if (KS != 1 && new_chunk) {          // every wave is past its reads of the (single) patch buffer: split + store the prefetched chunk
            X3_STORE_X(0)
            __syncthreads();
        }
        tap = ntap;
        chunk = nchunk;
    }
#undef X3_STAGE_X
#undef X3_LOAD_X
#undef X3_STORE_X
#undef X3_LOAD_W
#undef X3_STORE_W

#pragma unroll
    for (int pr = 0; pr < PX_T; ++pr) {
        const int oy = oy0 + wrow + pr, ox = ox0 + lrow;
        if (oy < Ho && ox < Wo) {
            const size_t pix = (size_t)oy * Wo + ox;
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = wch + ct * 32 + 8 * q + 4 * (lane >> 5);
                    const float4 sc = *reinterpret_cast<const float4 *>(SS + cl);
                    const float4 sh = *reinterpret_cast<const float4 *>(SS + BN + cl);
                    constexpr float inv = 1.0f / X3_SCALE;
                    float v0 = (accm[ct][pr][4 * q + 0] + accl[ct][pr][4 * q + 0] * inv) * sc.x + sh.x;
                    float v1 = (accm[ct][pr][4 * q + 1] + accl[ct][pr][4 * q + 1] * inv) * sc.y + sh.y;
                    float v2 = (accm[ct][pr][4 * q + 2] + accl[ct][pr][4 * q + 2] * inv) * sc.z + sh.z;
                    float v3 = (accm[ct][pr][4 * q + 3] + accl[ct][pr][4 * q + 3] * inv) * sc.w + sh.w;
                    const size_t o = pix * CoutP + n0 + cl;
                    if (HAS_RES) {
                        const float4 r = *reinterpret_cast<const float4 *>(res + o);
                        v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                    }
                    if (relu) {
                        v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                    }
                    *reinterpret_cast<float4 *>(out + o) = make_float4(v0, v1, v2, v3);
                }
        }
    }
}

// packed fp32 filters -> the same array with every float4 replaced by (4 hi halves, 4 lo halves)
__global__ __launch_bounds__(NT)
void x3_split_kernel(const float4 *__restrict__ w, size_t n4, uint4 *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n4) return;
    h4_t hi, lo;
    x3_split(w[i], hi, lo);
    uint4 o;
    __builtin_memcpy(&o.x, &hi, 8);
    __builtin_memcpy(&o.z, &lo, 8);
    out[i] = o;
}

// the same split into two PLANES of n halves each (hi, then lo'): the form conv3x3_pp's three-pass instantiation stages without
// arithmetic (activations [H][W][C] and packed filters alike)
__global__ __launch_bounds__(NT)
void x3_split_planes_kernel(const float4 *__restrict__ in, size_t n4, uint2 *__restrict__ hi_out, uint2 *__restrict__ lo_out)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n4) return;
    h4_t hi, lo;
    x3_split(in[i], hi, lo);
    uint2 a, b;
    __builtin_memcpy(&a, &hi, 8);
    __builtin_memcpy(&b, &lo, 8);
    hi_out[i] = a;
    lo_out[i] = b;
}

void launch_x3_split_planes(hipStream_t st, const float *in, size_t n_floats, void *hi_out, void *lo_out)
{
    const size_t n4 = n_floats / 4;
    hipLaunchKernelGGL(x3_split_planes_kernel, dim3((unsigned)((n4 + NT - 1) / NT)), dim3(NT), 0, st,
                       reinterpret_cast<const float4 *>(in), n4, reinterpret_cast<uint2 *>(hi_out), reinterpret_cast<uint2 *>(lo_out));
}

void launch_x3_split(hipStream_t st, const float *w, size_t n_floats, void *out)
{
    const size_t n4 = n_floats / 4;
    hipLaunchKernelGGL(x3_split_kernel, dim3((unsigned)((n4 + NT - 1) / NT)), dim3(NT), 0, st,
                       reinterpret_cast<const float4 *>(w), n4, reinterpret_cast<uint4 *>(out));
}

template <int KS, int STRIDE, int BN, bool HAS_RES>
static void launch_x3_t(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk, const float *scale,
                        const float *shift, int CoutP, int relu, const float *res, float *out, int Ho, int Wo)
{
    constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    constexpr size_t lds = (size_t)((KS == 1 ? 2 : 1) * PH * PW + 2 * BN) * PIXH * sizeof(half_t) * 2 + (size_t)2 * BN * sizeof(float);
    static bool attr_done = false;
    auto kern = conv_igemm_x3_kernel<KS, STRIDE, BN, HAS_RES>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    hipLaunchKernelGGL(kern, dim3(tiles_x * tiles_y * (CoutP / BN)), dim3(NT), lds, st, in, H, W, Cin, wpk, scale, shift,
                       CoutP, relu, res, out, Ho, Wo, tiles_x);
}

void launch_conv_igemm_x3(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk,
                          const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                          const float *residual, float *out, int Ho, int Wo)
{
#define SFD2_X3(KS_, ST_, BN_)                                                                                         \
    do {                                                                                                              \
        if (residual) launch_x3_t<KS_, ST_, BN_, true>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo); \
        else launch_x3_t<KS_, ST_, BN_, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo);         \
    } while (0)
    const bool b128 = (CoutP % 128 == 0);
    if (ks == 3 && stride == 1) { if (b128) SFD2_X3(3, 1, 128); else SFD2_X3(3, 1, 64); }
    else if (ks == 3 && stride == 2) { if (b128) SFD2_X3(3, 2, 128); else SFD2_X3(3, 2, 64); }
    else if (ks == 1 && stride == 1) { if (b128) SFD2_X3(1, 1, 128); else SFD2_X3(1, 1, 64); }
    else abort();
#undef SFD2_X3
}

// ---------------------------------------------------------------------------------------------
// ResBlock.conv2 (3x3, groups = 32) in the f16x3 mode: gconv3x3_g8_kernel's layout (two groups per 16x16x32 MFMA with
// block-diagonal filter fragments, 5 k steps for the 9 taps) on fp32 activations, three passes as conv_igemm_x3_kernel.
// The 64-channel patch chunk is split into hi / lo planes while it is staged; the filter fragments are pre-split
// (gconv_x3_pack_kernel, once per context): [16 pairs][5 steps][64 lanes][8 hi | 8 lo].
#define GX_P 72   // halves per pixel record of one plane: 64 + 8 pad
#define GX_PH 6
#define GX_PW 34
__global__ __launch_bounds__(NT)
void gconv_x3_pack_kernel(const float *__restrict__ w /*[256][8][3][3]*/, half_t *__restrict__ out)
{
    const int idx = blockIdx.x * NT + threadIdx.x;          // (pair * 5 + s) * 64 + lane
    if (idx >= 16 * 5 * 64) return;
    const int lane = idx & 63, s5 = (idx >> 6) % 5, pair = idx / (5 * 64);
    const int i = lane & 15, g = lane >> 4;
    const int tap = 2 * s5 + (g >> 1), oc = pair * 16 + i;
    const bool live = tap <= 8 && (i >> 3) == (g & 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = live ? w[((size_t)oc * 8 + j) * 9 + tap] : 0.0f;
        const half_t hi = (half_t)v;
        out[(size_t)idx * 16 + j] = hi;
        out[(size_t)idx * 16 + 8 + j] = (half_t)((v - (float)hi) * X3_SCALE);
    }
}

void launch_gconv_x3_pack(hipStream_t st, const float *w, void *out)
{
    hipLaunchKernelGGL(gconv_x3_pack_kernel, dim3((16 * 5 * 64 + NT - 1) / NT), dim3(NT), 0, st, w, reinterpret_cast<half_t *>(out));
}

__global__ __launch_bounds__(NT)
void gconv_x3_kernel(const float *__restrict__ in, int H, int W, const half_t *__restrict__ wpk,
                     const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out, int tiles_x)
{
    constexpr int NPIX = GX_PH * GX_PW;
    __shared__ __attribute__((aligned(16))) half_t Xh[NPIX * GX_P];
    __shared__ __attribute__((aligned(16))) half_t Xl[NPIX * GX_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_swizzle_f(blockIdx.x, gridDim.x);
    const int tx = swz % tiles_x, ty = swz / tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int g = lane >> 4, lcol = lane & 15;

    constexpr int NLD = (NPIX * 16 + NT - 1) / NT;   // float4 pieces per thread per 64-channel chunk (13)
    float4 pre[NLD];
#define GX_FETCH(chunk_)                                                                                  \
    _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                                     \
        const int p = tid + k * NT;                                                                       \
        const int q = p >> 4, part = p & 15;                                                              \
        const int py = q / GX_PW, px = q - py * GX_PW;                                                    \
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;                                                   \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                       \
        if (p < NPIX * 16 && iy >= 0 && iy < H && ix >= 0 && ix < W)                                      \
            v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * 256 + (chunk_)*64 + part * 4)); \
        pre[k] = v;                                                                                       \
    }
    GX_FETCH(0)
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading the planes
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int p = tid + k * NT;
            if (p < NPIX * 16) {
                h4_t hi, lo;
                x3_split(pre[k], hi, lo);
                *reinterpret_cast<h4_t *>(Xh + (p >> 4) * GX_P + (p & 15) * 4) = hi;
                *reinterpret_cast<h4_t *>(Xl + (p >> 4) * GX_P + (p & 15) * 4) = lo;
            }
        }
        const int pair = chunk * 4 + wave;
        h8_t wh[5], wl[5];
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const half_t *wp = wpk + ((size_t)(pair * 5 + s5) * 64 + lane) * 16;
            wh[s5] = *reinterpret_cast<const h8_t *>(wp);
            wl[s5] = *reinterpret_cast<const h8_t *>(wp + 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // planes of this chunk complete
        if (chunk + 1 < 4) { GX_FETCH(chunk + 1) }

        f32x4_t accm[8], accl[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { accm[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; accl[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            int tap = 2 * s5 + (g >> 1);
            if (tap > 8) tap = 8;  // zero-weight slot: read any valid location
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = ((t >> 1) + ky) * GX_PW + (t & 1) * 16 + lcol + kx;
                const h8_t bh = *reinterpret_cast<const h8_t *>(Xh + q * GX_P + wave * 16 + (g & 1) * 8);
                const h8_t bl = *reinterpret_cast<const h8_t *>(Xl + q * GX_P + wave * 16 + (g & 1) * 8);
                accm[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s5], bh, accm[t], 0, 0, 0);
                accl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s5], bl, accl[t], 0, 0, 0);
                accl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s5], bh, accl[t], 0, 0, 0);
            }
        }
        const int c0 = pair * 16 + g * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
        constexpr float inv = 1.0f / X3_SCALE;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int oy = oy0 + (t >> 1), ox = ox0 + (t & 1) * 16 + lcol;
            if (oy < H && ox < W) {
                const float4 v = make_float4(fmaxf((accm[t][0] + accl[t][0] * inv) * sc.x + sh.x, 0.0f), fmaxf((accm[t][1] + accl[t][1] * inv) * sc.y + sh.y, 0.0f),
                                             fmaxf((accm[t][2] + accl[t][2] * inv) * sc.z + sh.z, 0.0f), fmaxf((accm[t][3] + accl[t][3] * inv) * sc.w + sh.w, 0.0f));
                const size_t o = ((size_t)oy * W + ox) * 256 + c0;
                *reinterpret_cast<float4 *>(out + o) = v;
            }
        }
    }
#undef GX_FETCH
}

void launch_gconv_x3(hipStream_t st, const float *in, int H, int W, const void *wpk, const float *scale, const float *shift,
                     float *out)
{
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL(gconv_x3_kernel, dim3(tiles_x * tiles_y), dim3(NT), 0, st, in, H, W, reinterpret_cast<const half_t *>(wpk),
                       scale, shift, out, tiles_x);
}

// ---------------------------------------------------------------------------------------------
// conv1a (3 -> 64, 3x3) + norm_RGB + BN + ReLU, fp32 VALU, accumulation order (c, ky, kx) as the
// reference's direct convolution.  One thread = one pixel, 64 output channels in 4 passes of 16.
__global__ __launch_bounds__(NT)
void conv1a_f32_kernel(const float *__restrict__ img, int H, int W, int normalise, const float *__restrict__ w /*[64][27]*/,
                       const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out, int tiles_x)
{
    __shared__ float P[3][10][34];
    __shared__ float Wl[27][64];
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const size_t plane = (size_t)H * W;
    for (int p = tid; p < 3 * 10 * 34; p += NT) {
        const int c = p / 340, r = p - c * 340, py = r / 34, px = r - py * 34;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        float v = 0.0f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            if (normalise & 2) {  // uint8 HWC ingest (extract_localization.py:165-186)
                const int cs = (normalise & 4) ? 2 - c : c;
                v = __fdiv_rn((float)reinterpret_cast<const unsigned char *>(img)[((size_t)iy * W + ix) * 3 + cs], 255.0f);
            } else {
                v = img[c * plane + (size_t)iy * W + ix];
            }
            if (normalise & 1) {
                const float m = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
                const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
                v = __fdiv_rn(__fsub_rn(v, m), sd);
            }
        }
        P[c][py][px] = v;
    }
    for (int p = tid; p < 27 * 64; p += NT) {
        const int oc = p & 63, k = p >> 6;
        Wl[k][oc] = w[oc * 27 + k];
    }
    __syncthreads();
    const int py = tid >> 5, px = tid & 31;
    const int oy = oy0 + py, ox = ox0 + px;
    float x[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) x[c * 9 + ky * 3 + kx] = P[c][py + ky][px + kx];
    if (oy >= H || ox >= W) return;
    float *o = out + ((size_t)oy * W + ox) * 64;
#pragma unroll 1
    for (int cb = 0; cb < 64; cb += 16) {
        float a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = fmaf(Wl[k][cb + j], x[k], a[j]);
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            float4 v;
            v.x = fmaxf(a[j] * scale[cb + j] + shift[cb + j], 0.0f);
            v.y = fmaxf(a[j + 1] * scale[cb + j + 1] + shift[cb + j + 1], 0.0f);
            v.z = fmaxf(a[j + 2] * scale[cb + j + 2] + shift[cb + j + 2], 0.0f);
            v.w = fmaxf(a[j + 3] * scale[cb + j + 3] + shift[cb + j + 3], 0.0f);
            *reinterpret_cast<float4 *>(o + cb + j) = v;
        }
    }
}

void launch_conv1a_f32(hipStream_t st, const float *img, int H, int W, int normalise, const float *w,
                       const float *scale, const float *shift, float *out)
{
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    hipLaunchKernelGGL(conv1a_f32_kernel, dim3(tiles_x * tiles_y), dim3(NT), 0, st, img, H, W, normalise, w, scale, shift,
                       out, tiles_x);
}

// ---------------------------------------------------------------------------------------------
// ResBlock.conv2 (3x3, groups = 32, 8 channels per group) + BN + ReLU, fp32 VALU.
// Block: 4 x 32 pixels, 64 channels (8 groups) per pass; the patch is staged channel-major so a
// wave's lanes (consecutive x) read consecutive LDS words.
__global__ __launch_bounds__(NT)
void gconv_f32_kernel(const float *__restrict__ in, int H, int W, const float *__restrict__ w /*[256][8][3][3]*/,
                      const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out, int tiles_x)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *P = reinterpret_cast<float *>(smem);      // [64 ch][6][36]
    float *Wl = P + 64 * 6 * 36;                     // [64 oc][72]
    const int tid = threadIdx.x;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int oy0 = ty * 4, ox0 = tx * 32;
    const int pix = tid & 127, half = tid >> 7;      // half selects 4 of the 8 groups of the pass
    const int py = pix >> 5, px = pix & 31;
    const int oy = oy0 + py, ox = ox0 + px;
    for (int pass = 0; pass < 4; ++pass) {
        __syncthreads();
        for (int p = tid; p < 6 * 34 * 16; p += NT) {   // 16 float4 per pixel (64 channels)
            const int q = p >> 4, part = p & 15;
            const int yy = q / 34, xx = q - yy * 34;
            const int iy = oy0 - 1 + yy, ix = ox0 - 1 + xx;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = *reinterpret_cast<const float4 *>(in + ((size_t)(iy * W + ix) * 256 + pass * 64 + part * 4));
            const int c = part * 4;
            P[((c + 0) * 6 + yy) * 36 + xx] = v.x;
            P[((c + 1) * 6 + yy) * 36 + xx] = v.y;
            P[((c + 2) * 6 + yy) * 36 + xx] = v.z;
            P[((c + 3) * 6 + yy) * 36 + xx] = v.w;
        }
        for (int p = tid; p < 64 * 72; p += NT) Wl[p] = w[(size_t)pass * 64 * 72 + p];
        __syncthreads();
#pragma unroll 1
        for (int gi = 0; gi < 4; ++gi) {
            const int g = half * 4 + gi;             // group within the pass
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = 0.0f;
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci) {
                float x[9];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) x[ky * 3 + kx] = P[((g * 8 + ci) * 6 + py + ky) * 36 + px + kx];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int t = 0; t < 9; ++t) a[j] = fmaf(Wl[(g * 8 + j) * 72 + ci * 9 + t], x[t], a[j]);
            }
            if (oy < H && ox < W) {
                const int c0 = pass * 64 + g * 8;
                float *o = out + ((size_t)oy * W + ox) * 256 + c0;
                float4 v0, v1;
                v0.x = fmaxf(a[0] * scale[c0] + shift[c0], 0.0f);
                v0.y = fmaxf(a[1] * scale[c0 + 1] + shift[c0 + 1], 0.0f);
                v0.z = fmaxf(a[2] * scale[c0 + 2] + shift[c0 + 2], 0.0f);
                v0.w = fmaxf(a[3] * scale[c0 + 3] + shift[c0 + 3], 0.0f);
                v1.x = fmaxf(a[4] * scale[c0 + 4] + shift[c0 + 4], 0.0f);
                v1.y = fmaxf(a[5] * scale[c0 + 5] + shift[c0 + 5], 0.0f);
                v1.z = fmaxf(a[6] * scale[c0 + 6] + shift[c0 + 6], 0.0f);
                v1.w = fmaxf(a[7] * scale[c0 + 7] + shift[c0 + 7], 0.0f);
                *reinterpret_cast<float4 *>(o) = v0;
                *reinterpret_cast<float4 *>(o + 4) = v1;
            }
        }
    }
}

void launch_gconv_f32(hipStream_t st, const float *in, int H, int W, const float *w, const float *scale,
                      const float *shift, float *out)
{
    static bool attr_done = false;
    const size_t lds = (size_t)(64 * 6 * 36 + 64 * 72) * sizeof(float);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gconv_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 3) / 4;
    hipLaunchKernelGGL(gconv_f32_kernel, dim3(tiles_x * tiles_y), dim3(NT), lds, st, in, H, W, w, scale, shift, out, tiles_x);
}

// ---------------------------------------------------------------------------------------------
// ConvSta on fp32 activations (16 lanes per pixel, as convsta_kernel)
__global__ __launch_bounds__(NT)
void convsta_f32_kernel(const float *__restrict__ in, int npix, const float *__restrict__ w, const float *__restrict__ b,
                        float *__restrict__ out)
{
    const int l16 = threadIdx.x & 15;
    float wr[3][16];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
        for (int j = 0; j < 16; ++j) wr[o][j] = w[o * 256 + l16 * 16 + j];
    const float b0 = b[0], b1 = b[1], b2 = b[2];
    const int gstride = (gridDim.x * blockDim.x) >> 4;
    for (int pix = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; pix < npix; pix += gstride) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = *reinterpret_cast<const float4 *>(in + (size_t)pix * 256 + l16 * 16 + q * 4);
            s0 += x.x * wr[0][4 * q] + x.y * wr[0][4 * q + 1] + x.z * wr[0][4 * q + 2] + x.w * wr[0][4 * q + 3];
            s1 += x.x * wr[1][4 * q] + x.y * wr[1][4 * q + 1] + x.z * wr[1][4 * q + 2] + x.w * wr[1][4 * q + 3];
            s2 += x.x * wr[2][4 * q] + x.y * wr[2][4 * q + 1] + x.z * wr[2][4 * q + 2] + x.w * wr[2][4 * q + 3];
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            s0 += __shfl_xor(s0, m);
            s1 += __shfl_xor(s1, m);
            s2 += __shfl_xor(s2, m);
        }
        if (l16 == 0) {
            out[pix] = s0 + b0;
            out[(size_t)npix + pix] = s1 + b1;
            out[2 * (size_t)npix + pix] = s2 + b2;
        }
    }
}

void launch_convsta_f32(hipStream_t st, const float *in, int npix, const float *w, const float *b, float *out)
{
    int grid = (npix * 16 + NT - 1) / NT;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(convsta_f32_kernel, dim3(grid), dim3(NT), 0, st, in, npix, w, b, out);
}
