// ResSegNetV2 conv stack for gfx950 (MI355X): implicit-GEMM convolutions on
// v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16 with the conv bias + BatchNorm(eval)
// (+ residual) (+ ReLU) folded into the epilogue.
//
// Replaces the torch.nn.Conv2d / BatchNorm2d / ReLU modules of nets/sfd2.py:259-303
// as executed by ResSegNetV2.det (nets/sfd2.py:313-326, :328, :340, :345).
//
// Data layout in HBM: activations NHWC fp16 (channels innermost, so the GEMM K
// dimension of every filter tap is one contiguous 64..512-byte run per pixel);
// filters pre-packed per (32-channel K chunk, tap) as [Cout][32] fp16.
//
// GEMM orientation: MFMA "A" rows = output channels (filters), "B" columns = output
// pixels.  With the 32x32 C/D layout (col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5))
// a lane then owns 4 consecutive output channels of one pixel per register quad, i.e.
// 8-byte NHWC stores.
#include "sfd2_internal.h"
#include <stdlib.h>

#define TW 32     // output tile width  (one 32-wide MFMA column block = 32 consecutive pixels of a row)
#define TH 4      // output tile height
#define CC 32     // input channels per K chunk
#define PIXP 40   // fp16 elements per pixel record in LDS: 32 + 8 pad (80 B: conflict-free ds_read_b128)
#define NT 256

__device__ __forceinline__ int xcd_swizzle(int bid, int nblk)
{
    // Blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8).  Give every XCD one
    // contiguous run of tiles so that halo rows/columns of neighbouring tiles hit in its L2.
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ h4_t cvt4(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// ---------------------------------------------------------------------------------------------
// Generic implicit-GEMM conv.  Block = 256 threads = 4 waves, output tile 4 x 32 pixels x BN
// channels.  K loop: for each 32-channel chunk the input patch (with halo) is staged once in LDS
// and re-used by all KS*KS taps; one [BN][32] filter tile is staged per (chunk, tap) step.
// Both are double-buffered through registers (global loads issued before the MFMAs of the
// current step, LDS stores after), one barrier per step.
template <int KS, int STRIDE, int BN, bool OUT_F32, bool HAS_RES>
__global__ __launch_bounds__(NT, (STRIDE == 1 ? 2 : 1))
void conv_igemm_kernel(const half_t *__restrict__ in, int H, int W, int Cin,
                       const half_t *__restrict__ wpk, const float *__restrict__ scale,
                       const float *__restrict__ shift, int CoutP, int relu,
                       const half_t *__restrict__ res, void *__restrict__ outv,
                       int Ho, int Wo, int tiles_x)
{
    constexpr int T = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int PH = (TH - 1) * STRIDE + KS;
    constexpr int PW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = PH * PW;
    constexpr int XPIECES = NPIX * 4;                  // 16-byte pieces of one patch chunk
    constexpr int XP = (XPIECES + NT - 1) / NT;
    constexpr int WP = BN * 4 / NT;
    constexpr int WAVES_CH = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_PX = 4 / WAVES_CH;
    constexpr int CH_T = BN / WAVES_CH / 32;           // 32-channel MFMA row blocks per wave
    constexpr int PX_T = TH / WAVES_PX;                // image rows (32-pixel column blocks) per wave

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t *Xs = reinterpret_cast<half_t *>(smem);     // [2][NPIX][PIXP]
    half_t *Ws = Xs + 2 * NPIX * PIXP;                 // [2][BN][PIXP]
    float *SS = reinterpret_cast<float *>(Ws + 2 * BN * PIXP);   // scale[BN], shift[BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wch = (wave % WAVES_CH) * (CH_T * 32);
    const int wrow = (wave / WAVES_CH) * PX_T;

    const int n_tiles_n = CoutP / BN;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tn = swz % n_tiles_n;
    const int tsp = swz / n_tiles_n;
    const int tx = tsp % tiles_x, ty = tsp / tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;

    uint4 xr[XP], wr[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = make_uint4(0, 0, 0, 0);

// Staging helpers are macros (not lambdas): by-reference captures kept the register arrays in scratch.
#define LOAD_X(chunk_)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                          \
        const int p = tid + i * NT;                                                                            \
        uint4 v = make_uint4(0, 0, 0, 0);                                                                      \
        if (p < XPIECES) {                                                                                     \
            const int q = p >> 2, part = p & 3;                                                                \
            const int py = q / PW, px = q - py * PW;                                                           \
            const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                              \
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)                                                        \
                v = *reinterpret_cast<const uint4 *>(in + ((size_t)(iy * W + ix) * Cin + (chunk_)*CC + part * 8)); \
        }                                                                                                      \
        xr[i] = v;                                                                                             \
    }
#define STORE_X(buf_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                          \
        const int p = tid + i * NT;                                                                            \
        if (p < XPIECES) {                                                                                     \
            const int q = p >> 2, part = p & 3;                                                                \
            *reinterpret_cast<uint4 *>(Xs + ((buf_)*NPIX + q) * PIXP + part * 8) = xr[i];                      \
        }                                                                                                      \
    }
#define LOAD_W(step_)                                                                                         \
    {                                                                                                          \
        const half_t *wbase_ = wpk + ((size_t)(step_)*CoutP + n0) * CC;                                        \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                                         \
            wr[i] = *reinterpret_cast<const uint4 *>(wbase_ + (size_t)(tid + i * NT) * 8);                     \
    }
#define STORE_W(buf_)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                          \
        const int p = tid + i * NT, row = p >> 2, part = p & 3;                                                \
        *reinterpret_cast<uint4 *>(Ws + ((buf_)*BN + row) * PIXP + part * 8) = wr[i];                          \
    }

    f32x16_t acc[CH_T][PX_T];
#pragma unroll
    for (int a = 0; a < CH_T; ++a)
#pragma unroll
        for (int b = 0; b < PX_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int NS = (Cin / CC) * T;
    LOAD_X(0)
    LOAD_W(0)
    STORE_X(0)
    STORE_W(0)
    if (tid < BN) { SS[tid] = scale[n0 + tid]; SS[BN + tid] = shift[n0 + tid]; }
    __syncthreads();

    const int lrow = lane & 31, lk = (lane >> 5) * 8;
    int chunk = 0, tap = 0;
    for (int s = 0; s < NS; ++s) {
        const int wb = s & 1, xb = chunk & 1;
        int ntap = tap + 1, nchunk = chunk;
        if (ntap == T) { ntap = 0; ++nchunk; }
        const bool has_next = (s + 1 < NS);
        const bool new_chunk = has_next && (ntap == 0);
        if (has_next) LOAD_W(s + 1)
        if (new_chunk) { LOAD_X(nchunk) }

        const int ky = tap / KS, kx = tap - ky * KS;
        const half_t *xs = Xs + xb * NPIX * PIXP;
        const half_t *ws = Ws + wb * BN * PIXP;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8_t a[CH_T], b[PX_T];
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
                a[ct] = *reinterpret_cast<const h8_t *>(ws + (wch + ct * 32 + lrow) * PIXP + kk * 16 + lk);
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr) {
                const int q = ((wrow + pr) * STRIDE + ky) * PW + lrow * STRIDE + kx;
                b[pr] = *reinterpret_cast<const h8_t *>(xs + q * PIXP + kk * 16 + lk);
            }
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int pr = 0; pr < PX_T; ++pr)
                    acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ct], b[pr], acc[ct][pr], 0, 0, 0);
        }
        if (has_next) { STORE_W(wb ^ 1) }
        if (new_chunk) { STORE_X(xb ^ 1) }
        __syncthreads();
        tap = ntap;
        chunk = nchunk;
    }

#undef LOAD_X
#undef STORE_X
#undef LOAD_W
#undef STORE_W

    // epilogue: y = acc * scale[c] + shift[c] (+ residual) (ReLU); lane owns pixel (lane&31) and, per
    // register quad q, channels 8q + 4*(lane>>5) .. +3 of each 32-channel block.  scale/shift come
    // from LDS; the residual quads of one block are fetched together before use.
#pragma unroll
    for (int pr = 0; pr < PX_T; ++pr) {
        const int oy = oy0 + wrow + pr, ox = ox0 + lrow;
        if (oy < Ho && ox < Wo) {
            const size_t pix = (size_t)oy * Wo + ox;
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct) {
                const int cl = wch + ct * 32 + 4 * (lane >> 5);
                h4_t rq[4];
                if (HAS_RES) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        rq[q] = *reinterpret_cast<const h4_t *>(res + pix * CoutP + n0 + cl + 8 * q);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 sc = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                    const float4 sh = *reinterpret_cast<const float4 *>(SS + BN + cl + 8 * q);
                    float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                    float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                    float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                    float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                    if (HAS_RES) {
                        v0 += (float)rq[q][0]; v1 += (float)rq[q][1]; v2 += (float)rq[q][2]; v3 += (float)rq[q][3];
                    }
                    if (relu) {
                        v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                    }
                    const size_t o = pix * CoutP + n0 + cl + 8 * q;
                    if (OUT_F32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + o) = make_float4(v0, v1, v2, v3);
                    else *reinterpret_cast<h4_t *>(reinterpret_cast<half_t *>(outv) + o) = cvt4(v0, v1, v2, v3);
                }
            }
        }
    }
}

template <int KS, int STRIDE, int BN, bool OUT_F32, bool HAS_RES>
static void launch_igemm_t(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                           const float *scale, const float *shift, int CoutP, int relu, const half_t *res,
                           void *out, int Ho, int Wo)
{
    constexpr int PH = (TH - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    constexpr size_t lds = (size_t)(2 * PH * PW + 2 * BN) * PIXP * sizeof(half_t) + (size_t)2 * BN * sizeof(float);
    static bool attr_done = false;
    auto kern = conv_igemm_kernel<KS, STRIDE, BN, OUT_F32, HAS_RES>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    const int grid = tiles_x * tiles_y * (CoutP / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, res, out,
                       Ho, Wo, tiles_x);
}

bool launch_conv_igemm2(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                        const half_t *residual, void *out, int out_f32, int Ho, int Wo, const half_t *zero_page);
int conv_igemm2_chunk(int ks, int stride, int CoutP, int Cin);

// K-chunk width (input channels per packed filter tile) of the kernel that will run this layer.
// SFD2_CONV_V1=1 forces the first-generation kernel (32-wide chunks) everywhere.
int conv_igemm_chunk(int ks, int stride, int CoutP, int Cin)
{
    static const bool force_v1 = sfd2_env("SFD2_CONV_V1") != nullptr;
    if (!force_v1) {
        const int c = conv_igemm2_chunk(ks, stride, CoutP, Cin);
        if (c) return c;
    }
    return 32;
}

void launch_conv_igemm(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                       const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                       const half_t *residual, void *out, int out_f32, int Ho, int Wo, const half_t *zero_page)
{
    // second-generation kernel (conv2_kernels.hip) for the stride-1 layers; SFD2_CONV_V1=1 forces the first
    if (sfd2_env("SFD2_CONV_V1") == nullptr && zero_page &&
        launch_conv_igemm2(st, in, H, W, Cin, wpk, scale, shift, CoutP, ks, stride, relu, residual, out, out_f32, Ho, Wo, zero_page))
        return;
#define SFD2_IGEMM(KS_, ST_, BN_, F32_)                                                                               \
    do {                                                                                                              \
        if (residual) launch_igemm_t<KS_, ST_, BN_, F32_, true>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo); \
        else launch_igemm_t<KS_, ST_, BN_, F32_, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo);         \
    } while (0)
    const int bn = (CoutP % 256 == 0) ? 256 : (CoutP % 128 == 0 ? 128 : 64);
    if (ks == 3 && stride == 1 && !out_f32) {
        if (bn == 256) SFD2_IGEMM(3, 1, 256, false);
        else if (bn == 128) SFD2_IGEMM(3, 1, 128, false);
        else SFD2_IGEMM(3, 1, 64, false);
    } else if (ks == 3 && stride == 2 && !out_f32) {
        if (bn == 256) SFD2_IGEMM(3, 2, 256, false);
        else if (bn == 128) SFD2_IGEMM(3, 2, 128, false);
        else SFD2_IGEMM(3, 2, 64, false);
    } else if (ks == 1 && stride == 1 && !out_f32) {
        if (bn == 256) SFD2_IGEMM(1, 1, 256, false);
        else if (bn == 128) SFD2_IGEMM(1, 1, 128, false);
        else SFD2_IGEMM(1, 1, 64, false);
    } else if (ks == 1 && stride == 1 && out_f32) {
        if (bn == 256) SFD2_IGEMM(1, 1, 256, true);
        else if (bn == 128) SFD2_IGEMM(1, 1, 128, true);
        else SFD2_IGEMM(1, 1, 64, true);
    } else {
        abort();  // no such layer on the SFD2 path
    }
#undef SFD2_IGEMM
}

// ---------------------------------------------------------------------------------------------
// conv1a: 3 -> 64, 3x3, stride 1 (nets/sfd2.py:268) with norm_RGB (nets/extractor.py:14-17,104)
// applied while the patch is staged.  K is laid out as k = ky*16 + kx*4 + c  (kx < 4, c < 4; the
// kx == 3 and c == 3 slots carry zero weights), i.e. one 32x32x16 MFMA per filter row.
#define C1_TH 8
#define C1_PH 10
#define C1_PW 36
__global__ __launch_bounds__(NT)
void conv1a_kernel(const float *__restrict__ img, int H, int W, int normalise,
                   const half_t *__restrict__ wpk, const float *__restrict__ scale,
                   const float *__restrict__ shift, half_t *__restrict__ out, int tiles_x)
{
    __shared__ __attribute__((aligned(16))) half_t Xs[C1_PH * C1_PW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = swz % tiles_x, ty = swz / tiles_x;
    const int oy0 = ty * C1_TH, ox0 = tx * TW;
    const size_t plane = (size_t)H * W;

    for (int p = tid; p < C1_PH * C1_PW; p += NT) {
        const int py = p / C1_PW, px = p - py * C1_PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const size_t o = (size_t)iy * W + ix;
            if (normalise & 2) {  // uint8 HWC ingest: x.astype(float32) / 255. (extract_localization.py:168,186)
                const unsigned char *u = reinterpret_cast<const unsigned char *>(img) + o * 3;
                const int sw = (normalise & 4) ? 2 : 0;  // BGR -> RGB (:165)
                r = __fdiv_rn((float)u[sw], 255.0f); g = __fdiv_rn((float)u[1], 255.0f); b = __fdiv_rn((float)u[2 - sw], 255.0f);
            } else {
                r = img[o]; g = img[plane + o]; b = img[2 * plane + o];
            }
            if (normalise & 1) {  // (x - mean) / std, one IEEE sub + one IEEE div as torchvision Normalize
                r = __fdiv_rn(__fsub_rn(r, 0.485f), 0.229f);
                g = __fdiv_rn(__fsub_rn(g, 0.456f), 0.224f);
                b = __fdiv_rn(__fsub_rn(b, 0.406f), 0.225f);
            }
        }
        *reinterpret_cast<h4_t *>(Xs + p * 4) = cvt4(r, g, b, 0.0f);
    }

    h8_t a[2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            a[ct][ky] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(ct * 3 + ky) * 64 + lane) * 8);
    __syncthreads();

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lrow = lane & 31, lg = lane >> 5;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const int q = (wave * 2 + pr + ky) * C1_PW + lrow + 2 * lg;
            const h4_t lo = *reinterpret_cast<const h4_t *>(Xs + q * 4);
            const h4_t hi = *reinterpret_cast<const h4_t *>(Xs + (q + 1) * 4);
            h8_t b;
            b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
            b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ct][ky], b, acc[ct][pr], 0, 0, 0);
        }

#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        const int oy = oy0 + wave * 2 + pr, ox = ox0 + lrow;
        if (oy < H && ox < W) {
            const size_t pix = (size_t)oy * W + ox;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = ct * 32 + 8 * q + 4 * lg;
                    const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
                    const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
                    const float v0 = fmaxf(acc[ct][pr][4 * q + 0] * sc.x + sh.x, 0.0f);
                    const float v1 = fmaxf(acc[ct][pr][4 * q + 1] * sc.y + sh.y, 0.0f);
                    const float v2 = fmaxf(acc[ct][pr][4 * q + 2] * sc.z + sh.z, 0.0f);
                    const float v3 = fmaxf(acc[ct][pr][4 * q + 3] * sc.w + sh.w, 0.0f);
                    *reinterpret_cast<h4_t *>(out + pix * 64 + c0) = cvt4(v0, v1, v2, v3);
                }
        }
    }
}

void launch_conv1a(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *wpk,
                   const float *scale, const float *shift, half_t *out)
{
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + C1_TH - 1) / C1_TH;
    hipLaunchKernelGGL(conv1a_kernel, dim3(tiles_x * tiles_y), dim3(NT), 0, st, img, H, W, normalise, wpk, scale,
                       shift, out, tiles_x);
}

// ---------------------------------------------------------------------------------------------
// ResBlock.conv2: 3x3, 256 -> 256, groups = 32 (8 channels per group)  (nets/sfd2.py:14-18, :32).
// Two groups form one 16x16x32 MFMA: rows = 16 output channels, K step = 2 taps x 16 input
// channels with a block-diagonal filter fragment (pre-packed per lane on the host), 5 steps
// cover the 9 taps (the 10th tap slot has zero weights).
#define GP 72  // fp16 per pixel record for a 64-channel chunk: 64 + 8 pad (144 B)
#define G_PH 6
#define G_PW 34
__global__ __launch_bounds__(NT)
void gconv3x3_g8_kernel(const half_t *__restrict__ in, int H, int W, const half_t *__restrict__ wpk,
                        const float *__restrict__ scale, const float *__restrict__ shift,
                        half_t *__restrict__ out, int tiles_x)
{
    constexpr int NPIX = G_PH * G_PW;
    __shared__ __attribute__((aligned(16))) half_t Xs[NPIX * GP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = swz % tiles_x, ty = swz / tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int g = lane >> 4, lcol = lane & 15;

    // The 64-channel patch of chunk c + 1 is fetched into registers while chunk c computes and stores; barriers are
    // LDS-only (s_barrier + lgkmcnt(0)) so neither that prefetch nor the output stores are drained at a barrier.
    constexpr int NLD = (NPIX * 8 + NT - 1) / NT;   // 16-byte pieces per thread (7)
    uint4 pre[NLD];
#define G_FETCH(chunk_)                                                                                   \
    _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                                     \
        const int p = tid + k * NT;                                                                       \
        const int q = p >> 3, part = p & 7;                                                               \
        const int py = q / G_PW, px = q - py * G_PW;                                                      \
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;                                                   \
        uint4 v = make_uint4(0, 0, 0, 0);                                                                 \
        if (p < NPIX * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)                                       \
            v = *reinterpret_cast<const uint4 *>(in + ((size_t)(iy * W + ix) * 256 + (chunk_)*64 + part * 8)); \
        pre[k] = v;                                                                                       \
    }
    G_FETCH(0)
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done reading Xs
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int p = tid + k * NT;
            if (p < NPIX * 8) *reinterpret_cast<uint4 *>(Xs + (p >> 3) * GP + (p & 7) * 8) = pre[k];
        }
        const int pair = chunk * 4 + wave;
        h8_t wf[5];
#pragma unroll
        for (int s = 0; s < 5; ++s)
            wf[s] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(pair * 5 + s) * 64 + lane) * 8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");               // Xs of this chunk complete
        if (chunk + 1 < 4) { G_FETCH(chunk + 1) }

        f32x4_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (f32x4_t){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            int tap = 2 * s + (g >> 1);
            if (tap > 8) tap = 8;  // zero-weight slot: read any valid location
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = ((t >> 1) + ky) * G_PW + (t & 1) * 16 + lcol + kx;
                const h8_t b = *reinterpret_cast<const h8_t *>(Xs + q * GP + wave * 16 + (g & 1) * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[s], b, acc[t], 0, 0, 0);
            }
        }
        const int c0 = pair * 16 + g * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
        // lane groups g and g^1 (16 lanes apart) hold adjacent 8-byte channel runs of the same pixel:
        // exchange across two pixel tiles so that every lane issues one 16-byte store per tile pair
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            uint2 pk[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const h4_t hv = cvt4(fmaxf(acc[t + j][0] * sc.x + sh.x, 0.0f), fmaxf(acc[t + j][1] * sc.y + sh.y, 0.0f),
                                     fmaxf(acc[t + j][2] * sc.z + sh.z, 0.0f), fmaxf(acc[t + j][3] * sc.w + sh.w, 0.0f));
                __builtin_memcpy(&pk[j], &hv, 8);
            }
            const bool odd = g & 1;
            const uint2 send = odd ? pk[0] : pk[1];
            uint2 recv;
            recv.x = __shfl_xor(send.x, 16);
            recv.y = __shfl_xor(send.y, 16);
            const int tt = odd ? t + 1 : t;                         // the tile this lane stores
            const int oy = oy0 + (tt >> 1), ox = ox0 + (tt & 1) * 16 + lcol;
            const uint4 v = odd ? make_uint4(recv.x, recv.y, pk[1].x, pk[1].y) : make_uint4(pk[0].x, pk[0].y, recv.x, recv.y);
            if (oy < H && ox < W)
                *reinterpret_cast<uint4 *>(out + ((size_t)oy * W + ox) * 256 + pair * 16 + (g & ~1) * 4) = v;
        }
    }
#undef G_FETCH
}

void launch_gconv3x3_g8(hipStream_t st, const half_t *in, int H, int W, const half_t *wpk, const float *scale,
                        const float *shift, half_t *out)
{
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    hipLaunchKernelGGL(gconv3x3_g8_kernel, dim3(tiles_x * tiles_y), dim3(NT), 0, st, in, H, W, wpk, scale, shift, out,
                       tiles_x);
}

// ---------------------------------------------------------------------------------------------
// ConvSta: 1x1, 256 -> 3 (nets/sfd2.py:303,345).  HBM-bound read of out4; 16 lanes per pixel.
__global__ __launch_bounds__(NT)
void convsta_kernel(const half_t *__restrict__ in, int npix, const float *__restrict__ w,
                    const float *__restrict__ b, float *__restrict__ out)
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
        const h8_t x0 = *reinterpret_cast<const h8_t *>(in + (size_t)pix * 256 + l16 * 16);
        const h8_t x1 = *reinterpret_cast<const h8_t *>(in + (size_t)pix * 256 + l16 * 16 + 8);
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)x0[j], c = (float)x1[j];
            s0 += a * wr[0][j] + c * wr[0][8 + j];
            s1 += a * wr[1][j] + c * wr[1][8 + j];
            s2 += a * wr[2][j] + c * wr[2][8 + j];
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

void launch_convsta(hipStream_t st, const half_t *in, int npix, const float *w, const float *b, float *out)
{
    int grid = (npix * 16 + NT - 1) / NT;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(convsta_kernel, dim3(grid), dim3(NT), 0, st, in, npix, w, b, out);
}
