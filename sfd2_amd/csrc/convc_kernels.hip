// Compensated-fp16 conv kernels (SFD2_PREC_F16C, see sfd2_internal.h): the generic layer kernel every compensated
// layer can run on, plus conv1a and the grouped ResBlock conv, whose operands do not have the generic shape.
//
//   convc_igemm_kernel   Conv2d (3x3 / 1x1, stride 1 / 2) + folded BN (+ residual) (+ ReLU) as an implicit GEMM: the
//                        fp16 chunks of the `hi` plane on v_mfma_f32_32x32x16_f16, then the same number of chunks of the
//                        `corr` plane on v_mfma_scale_f32_32x32x64_f8f6f4 against the corr filters (one instruction per
//                        tap and 32 channels), one fp32 accumulator set.  Register-staged (global -> VGPR -> LDS, 80-byte
//                        records), i.e. the schedule of the first-generation fp16 kernel: this is the reference
//                        implementation of the arithmetic that the tuned kernels (conv3x3_pp / conv3x3_rf / resblock in
//                        their COMP instantiations) are checked against.  nets/sfd2.py:58-95, :25-55.
//   conv1a_c_kernel      norm_RGB + conv1a (3 -> 64): image and filters split into fp16 hi + lo on the fly, three fp16
//                        MFMA passes (K is 27: the passes cost nothing), hi + corr planes out.  nets/sfd2.py:268.
//   gconv_c_kernel       ResBlock.conv2 (3x3, groups = 32): the corr plane's residual byte is widened to an fp16 `lo`
//                        while the patch is staged, three fp16 passes with hi / lo block-diagonal filter fragments.
//                        nets/sfd2.py:14-18, :32.
#include "sfd2_internal.h"
#include <stdlib.h>

#define CTW 32
#define CTH 4
#define CCC 32
#define CPIXP 40
#define CNT 256

__device__ __forceinline__ int xcd_swizzle_c(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// CIN_C: the input has a corr plane (in_c) and wpk holds 2 * Cin / 32 chunks; COUT_C: the corr plane of the output is
// written (out_c); residual: hi (+ corr when res_c != null).
template <int KS, int STRIDE, int BN, bool HAS_RES, bool CIN_C, bool COUT_C>
__global__ __launch_bounds__(CNT, (STRIDE == 1 ? 2 : 1))
void convc_igemm_kernel(const half_t *__restrict__ in, const half_t *__restrict__ in_c, int H, int W, int Cin,
                        const half_t *__restrict__ wpk, const float *__restrict__ scale,
                        const float *__restrict__ shift, int CoutP, int relu,
                        const half_t *__restrict__ res, const half_t *__restrict__ res_c,
                        half_t *__restrict__ out, half_t *__restrict__ out_c, int Ho, int Wo, int tiles_x, int sa,
                        unsigned int *__restrict__ range /* the output tensor's range-status slot, or null */)
{
    float mx = 0.0f;
    constexpr int T = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int PH = (CTH - 1) * STRIDE + KS;
    constexpr int PW = (CTW - 1) * STRIDE + KS;
    constexpr int NPIX = PH * PW;
    constexpr int XPIECES = NPIX * 4;
    constexpr int XP = (XPIECES + CNT - 1) / CNT;
    constexpr int WP = BN * 4 / CNT;
    constexpr int WAVES_CH = (BN >= 128) ? 2 : 1;
    constexpr int WAVES_PX = 4 / WAVES_CH;
    constexpr int CH_T = BN / WAVES_CH / 32;
    constexpr int PX_T = CTH / WAVES_PX;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half_t *Xs = reinterpret_cast<half_t *>(smem);     // [2][NPIX][CPIXP]
    half_t *Ws = Xs + 2 * NPIX * CPIXP;                // [2][BN][CPIXP]
    float *SS = reinterpret_cast<float *>(Ws + 2 * BN * CPIXP);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wch = (wave % WAVES_CH) * (CH_T * 32);
    const int wrow = (wave / WAVES_CH) * PX_T;

    const int n_tiles_n = CoutP / BN;
    const int swz = xcd_swizzle_c(blockIdx.x, gridDim.x);
    const int tn = swz % n_tiles_n;
    const int tsp = swz / n_tiles_n;
    const int tx = tsp % tiles_x, ty = tsp / tiles_x;
    const int oy0 = ty * CTH, ox0 = tx * CTW, n0 = tn * BN;
    const int NCH = Cin / CCC;                         // chunks per plane

    uint4 xr[XP], wr[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) xr[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = make_uint4(0, 0, 0, 0);

#define C_LOAD_X(chunk_)                                                                                      \
    {                                                                                                          \
        const half_t *pl_ = (CIN_C && (chunk_) >= NCH) ? in_c : in;                                            \
        const int cc_ = (CIN_C && (chunk_) >= NCH) ? (chunk_)-NCH : (chunk_);                                  \
        _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                       \
            const int p = tid + i * CNT;                                                                       \
            uint4 v = make_uint4(0, 0, 0, 0);                                                                  \
            if (p < XPIECES) {                                                                                 \
                const int q = p >> 2, part = p & 3;                                                            \
                const int py = q / PW, px = q - py * PW;                                                       \
                const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;                          \
                if (iy >= 0 && iy < H && ix >= 0 && ix < W)                                                    \
                    v = *reinterpret_cast<const uint4 *>(pl_ + ((size_t)(iy * W + ix) * Cin + cc_ * CCC + part * 8)); \
            }                                                                                                  \
            xr[i] = v;                                                                                         \
        }                                                                                                      \
    }
#define C_STORE_X(buf_)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < XP; ++i) {                                                          \
        const int p = tid + i * CNT;                                                                           \
        if (p < XPIECES) {                                                                                     \
            const int q = p >> 2, part = p & 3;                                                                \
            *reinterpret_cast<uint4 *>(Xs + ((buf_)*NPIX + q) * CPIXP + part * 8) = xr[i];                     \
        }                                                                                                      \
    }
#define C_LOAD_W(step_)                                                                                       \
    {                                                                                                          \
        const half_t *wbase_ = wpk + ((size_t)(step_)*CoutP + n0) * CCC;                                       \
        _Pragma("unroll") for (int i = 0; i < WP; ++i)                                                         \
            wr[i] = *reinterpret_cast<const uint4 *>(wbase_ + (size_t)(tid + i * CNT) * 8);                    \
    }
#define C_STORE_W(buf_)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < WP; ++i) {                                                          \
        const int p = tid + i * CNT, row = p >> 2, part = p & 3;                                               \
        *reinterpret_cast<uint4 *>(Ws + ((buf_)*BN + row) * CPIXP + part * 8) = wr[i];                         \
    }

    f32x16_t acc[CH_T][PX_T];
#pragma unroll
    for (int a = 0; a < CH_T; ++a)
#pragma unroll
        for (int b = 0; b < PX_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int NS = (CIN_C ? 2 : 1) * NCH * T;
    C_LOAD_X(0)
    C_LOAD_W(0)
    C_STORE_X(0)
    C_STORE_W(0)
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
        if (has_next) C_LOAD_W(s + 1)
        if (new_chunk) { C_LOAD_X(nchunk) }

        const int ky = tap / KS, kx = tap - ky * KS;
        const half_t *xs = Xs + xb * NPIX * CPIXP;
        const half_t *ws = Ws + wb * BN * CPIXP;
        h8_t a[2][CH_T], b[2][PX_T];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
                a[kk][ct] = *reinterpret_cast<const h8_t *>(ws + (wch + ct * 32 + lrow) * CPIXP + kk * 16 + lk);
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr) {
                const int q = ((wrow + pr) * STRIDE + ky) * PW + lrow * STRIDE + kx;
                b[kk][pr] = *reinterpret_cast<const h8_t *>(xs + q * CPIXP + kk * 16 + lk);
            }
        }
        if (CIN_C && chunk >= NCH) {      // block-uniform: the corr plane's chunks
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                for (int pr = 0; pr < PX_T; ++pr)
                    acc[ct][pr] = sfd2_mfma_corr(a[0][ct], a[1][ct], b[0][pr], b[1][pr], acc[ct][pr], sa);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                    for (int pr = 0; pr < PX_T; ++pr)
                        acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk][ct], b[kk][pr], acc[ct][pr], 0, 0, 0);
        }
        if (has_next) { C_STORE_W(wb ^ 1) }
        if (new_chunk) { C_STORE_X(xb ^ 1) }
        __syncthreads();
        tap = ntap;
        chunk = nchunk;
    }
#undef C_LOAD_X
#undef C_STORE_X
#undef C_LOAD_W
#undef C_STORE_W

    // epilogue: y = acc * scale + shift (+ residual hi + lo) (ReLU) -> hi plane (+ corr plane)
#pragma unroll
    for (int pr = 0; pr < PX_T; ++pr) {
        const int oy = oy0 + wrow + pr, ox = ox0 + lrow;
        if (oy < Ho && ox < Wo) {
            const size_t pix = (size_t)oy * Wo + ox;
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct) {
                const int cl = wch + ct * 32 + 4 * (lane >> 5);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const size_t o = pix * CoutP + n0 + cl + 8 * q;
                    const float4 sc = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                    const float4 sh = *reinterpret_cast<const float4 *>(SS + BN + cl + 8 * q);
                    float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                    float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                    float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                    float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                    if (HAS_RES) {
                        const h4_t rq = *reinterpret_cast<const h4_t *>(res + o);
                        v0 += (float)rq[0]; v1 += (float)rq[1]; v2 += (float)rq[2]; v3 += (float)rq[3];
                        if (res_c) {
                            const uint2 rc = *reinterpret_cast<const uint2 *>(res_c + o);
                            v0 += sfd2_corr_lo(rc.x, 0); v1 += sfd2_corr_lo(rc.x, 1);
                            v2 += sfd2_corr_lo(rc.y, 0); v3 += sfd2_corr_lo(rc.y, 1);
                        }
                    }
                    mx = sfd2_max3(mx, v0, v1);
                    mx = sfd2_max3(mx, v2, v3);
                    {   // ReLU and the saturation of compensated tensors at +-SFD2_C_SAT in one (as sfd2_epi4 in the tuned kernels)
                        const float lo = relu ? 0.0f : -SFD2_C_SAT;
                        v0 = __builtin_amdgcn_fmed3f(v0, lo, SFD2_C_SAT); v1 = __builtin_amdgcn_fmed3f(v1, lo, SFD2_C_SAT);
                        v2 = __builtin_amdgcn_fmed3f(v2, lo, SFD2_C_SAT); v3 = __builtin_amdgcn_fmed3f(v3, lo, SFD2_C_SAT);
                    }
                    uint2 hv, cv;
                    sfd2_split4(v0, v1, v2, v3, hv, cv);
                    *reinterpret_cast<uint2 *>(out + o) = hv;
                    if (COUT_C) *reinterpret_cast<uint2 *>(out_c + o) = cv;
                }
            }
        }
    }
    sfd2_range_commit(range, sfd2_wave_max_bits(mx));
}

template <int KS, int STRIDE, int BN, bool HAS_RES, bool CIN_C, bool COUT_C>
static void launch_convc_t(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                           const float *scale, const float *shift, int CoutP, int relu, const half_t *res, const half_t *res_c,
                           half_t *out, half_t *out_c, int Ho, int Wo, int sa, unsigned int *range)
{
    constexpr int PH = (CTH - 1) * STRIDE + KS, PW = (CTW - 1) * STRIDE + KS;
    constexpr size_t lds = (size_t)(2 * PH * PW + 2 * BN) * CPIXP * sizeof(half_t) + (size_t)2 * BN * sizeof(float);
    static bool attr_done = false;
    auto kern = convc_igemm_kernel<KS, STRIDE, BN, HAS_RES, CIN_C, COUT_C>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (Wo + CTW - 1) / CTW, tiles_y = (Ho + CTH - 1) / CTH;
    const int grid = tiles_x * tiles_y * (CoutP / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CNT), lds, st, in, in_c, H, W, Cin, wpk, scale, shift, CoutP, relu, res, res_c,
                       out, out_c, Ho, Wo, tiles_x, sa, range);
}

// Generic compensated layer.  wpk: [2 * Cin / 32][ks * ks][CoutP][32] units when in_c != null (fp16 chunks, then corr
// chunks), the first half only otherwise; sbyte: 127 - 9 - b0 of the layer.
void launch_convc_igemm(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                        const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, int Ho, int Wo, int sbyte, unsigned int *range)
{
    const int sa = (sbyte & 255) * 0x01010101;
    const int bn = (CoutP % 128 == 0) ? 128 : 64;   // 256-channel tiles spill here (two accumulator-heavy paths in one body)
#define CC_GO(KS_, ST_, BN_, RES_)                                                                                       \
    do {                                                                                                                \
        if (in_c && out_c) launch_convc_t<KS_, ST_, BN_, RES_, true, true>(st, in, in_c, H, W, Cin, wpk, scale, shift, CoutP, relu, res, res_c, out, out_c, Ho, Wo, sa, range); \
        else if (in_c) launch_convc_t<KS_, ST_, BN_, RES_, true, false>(st, in, in_c, H, W, Cin, wpk, scale, shift, CoutP, relu, res, res_c, out, out_c, Ho, Wo, sa, range);   \
        else launch_convc_t<KS_, ST_, BN_, RES_, false, true>(st, in, in_c, H, W, Cin, wpk, scale, shift, CoutP, relu, res, res_c, out, out_c, Ho, Wo, sa, range);             \
    } while (0)
#define CC_BN(KS_, ST_, RES_)                                                                                            \
    do {                                                                                                                \
        if (bn == 128) CC_GO(KS_, ST_, 128, RES_);                                                                 \
        else CC_GO(KS_, ST_, 64, RES_);                                                                                 \
    } while (0)
    if (ks == 3 && stride == 1 && !res) CC_BN(3, 1, false);
    else if (ks == 3 && stride == 2 && !res) CC_BN(3, 2, false);
    else if (ks == 1 && stride == 1 && res) CC_BN(1, 1, true);
    else if (ks == 1 && stride == 1) CC_BN(1, 1, false);
    else abort();   // no such layer on the SFD2 path
#undef CC_BN
#undef CC_GO
}

// ---------------------------------------------------------------------------------------------
// conv1a, compensated: K = ky * 16 + kx * 4 + c as in conv1a_kernel; image value x = xh + xl and filter w = wh + wl
// (all fp16; nothing here is small enough to leave fp16's normal range after the split), acc += wh xh + wl xh + wh xl.
#define C1C_TH 8
#define C1C_PH 10
#define C1C_PW 36
__global__ __launch_bounds__(CNT)
void conv1a_c_kernel(const float *__restrict__ img, int H, int W, int normalise,
                     const half_t *__restrict__ wpk /*[2 hi/lo][2][3][64][8]*/, const float *__restrict__ scale,
                     const float *__restrict__ shift, half_t *__restrict__ out, half_t *__restrict__ out_c, int tiles_x,
                     unsigned int *__restrict__ range)
{
    float mx = 0.0f;
    __shared__ __attribute__((aligned(16))) half_t Xh[C1C_PH * C1C_PW * 4];
    __shared__ __attribute__((aligned(16))) half_t Xl[C1C_PH * C1C_PW * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_swizzle_c(blockIdx.x, gridDim.x);
    const int tx = swz % tiles_x, ty = swz / tiles_x;
    const int oy0 = ty * C1C_TH, ox0 = tx * CTW;
    const size_t plane = (size_t)H * W;

    for (int p = tid; p < C1C_PH * C1C_PW; p += CNT) {
        const int py = p / C1C_PW, px = p - py * C1C_PW;
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
        h4_t hh, hl;
        hh[0] = (half_t)r; hh[1] = (half_t)g; hh[2] = (half_t)b; hh[3] = (half_t)0.0f;
        hl[0] = (half_t)(r - (float)hh[0]); hl[1] = (half_t)(g - (float)hh[1]); hl[2] = (half_t)(b - (float)hh[2]); hl[3] = (half_t)0.0f;
        *reinterpret_cast<h4_t *>(Xh + p * 4) = hh;
        *reinterpret_cast<h4_t *>(Xl + p * 4) = hl;
    }

    h8_t ah[2][3], al[2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            ah[ct][ky] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(ct * 3 + ky) * 64 + lane) * 8);
            al[ct][ky] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(6 + ct * 3 + ky) * 64 + lane) * 8);
        }
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
            const int q = (wave * 2 + pr + ky) * C1C_PW + lrow + 2 * lg;
            h8_t bh, bl;
            {
                const h4_t lo = *reinterpret_cast<const h4_t *>(Xh + q * 4);
                const h4_t hi = *reinterpret_cast<const h4_t *>(Xh + (q + 1) * 4);
                bh[0] = lo[0]; bh[1] = lo[1]; bh[2] = lo[2]; bh[3] = lo[3];
                bh[4] = hi[0]; bh[5] = hi[1]; bh[6] = hi[2]; bh[7] = hi[3];
            }
            {
                const h4_t lo = *reinterpret_cast<const h4_t *>(Xl + q * 4);
                const h4_t hi = *reinterpret_cast<const h4_t *>(Xl + (q + 1) * 4);
                bl[0] = lo[0]; bl[1] = lo[1]; bl[2] = lo[2]; bl[3] = lo[3];
                bl[4] = hi[0]; bl[5] = hi[1]; bl[6] = hi[2]; bl[7] = hi[3];
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ct][ky], bh, acc[ct][pr], 0, 0, 0);
                acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct][ky], bl, acc[ct][pr], 0, 0, 0);
                acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ct][ky], bh, acc[ct][pr], 0, 0, 0);
            }
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
                    const float y0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x, y1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                    const float y2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z, y3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                    mx = sfd2_max3(mx, y0, y1);
                    mx = sfd2_max3(mx, y2, y3);
                    const float v0 = __builtin_amdgcn_fmed3f(y0, 0.0f, SFD2_C_SAT), v1 = __builtin_amdgcn_fmed3f(y1, 0.0f, SFD2_C_SAT);
                    const float v2 = __builtin_amdgcn_fmed3f(y2, 0.0f, SFD2_C_SAT), v3 = __builtin_amdgcn_fmed3f(y3, 0.0f, SFD2_C_SAT);
                    uint2 hv, cv;
                    sfd2_split4(v0, v1, v2, v3, hv, cv);
                    *reinterpret_cast<uint2 *>(out + pix * 64 + c0) = hv;
                    *reinterpret_cast<uint2 *>(out_c + pix * 64 + c0) = cv;
                }
        }
    }
    sfd2_range_commit(range, sfd2_wave_max_bits(mx));
}

void launch_conv1a_c(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *wpk, const float *scale,
                     const float *shift, half_t *out, half_t *out_c, unsigned int *range)
{
    const int tiles_x = (W + CTW - 1) / CTW, tiles_y = (H + C1C_TH - 1) / C1C_TH;
    hipLaunchKernelGGL(conv1a_c_kernel, dim3(tiles_x * tiles_y), dim3(CNT), 0, st, img, H, W, normalise, wpk, scale, shift,
                       out, out_c, tiles_x, range);
}

// ---------------------------------------------------------------------------------------------
// ResBlock.conv2 (3x3, 256 -> 256, groups = 32), compensated.  Same 16x16 block-diagonal formulation as
// gconv3x3_g8_kernel for the hi plane (two groups = one 16-channel pair per MFMA row block, K step = 2 taps x 16 input
// channels on v_mfma_f32_16x16x32_f16, 5 steps); the corr plane goes to v_mfma_scale_f32_16x16x128_f8f6f4: a lane's 32
// bytes are the 16 corr units of the pair at ONE tap (contiguous in the pixel's record), the four lane groups are four
// taps, so 3 steps cover the 9 taps (12 slots, the last three with zero filters).  No conversion anywhere: both planes are
// staged as they are.  The layer moves 246 MB for 4 GFLOP: what it costs is the traffic.
#define GCP 72
#define GC_PH 6
#define GC_PW 34
typedef float f32x4c_t __attribute__((ext_vector_type(4)));
// IN_C = false: plain fp16 input (option "rb_inner" = 2: ResBlock.conv1 wrote only a hi plane) -- no corr records; the filter
// residuals arrive as fp16 fragments (w - fp16(w)) * 2^11 in wpk's layout (`wck`) for a second fp16 pass into its own
// accumulators.  OUT_C = false: only the hi plane is written ("rb_inner" >= 1).
// X3 (SFD2_PREC_F16X3 on the throughput path; IN_C = OUT_C = true): in_c / out_c are lo' planes (lo' = fp16((x - hi) * 2^11)) and wck
// the filters' lo' fragments: hi x hi into acc, hi x lo' + lo' x hi into acl, y = acc + acl * 2^-11 -- gconv_x3_kernel's arithmetic on
// pre-split operands.
template <bool IN_C, bool OUT_C, bool X3 = false>
__global__ __launch_bounds__(CNT, 2)   // two blocks (59 KB of LDS each) per CU
void gconv_c_kernel(const half_t *__restrict__ in, const half_t *__restrict__ in_c, int H, int W,
                    const half_t *__restrict__ wpk /*[16 pairs][5 steps][64 lanes][8] fp16*/,
                    const unsigned char *__restrict__ wck /*[16 pairs][3 steps][64 lanes][32 B] corr units*/,
                    const float *__restrict__ scale, const float *__restrict__ shift, half_t *__restrict__ out,
                    half_t *__restrict__ out_c, int tiles_x, int sa, int row0, int row1 /* output rows [row0, row1) of the image */,
                    unsigned int *__restrict__ range)
{
    float mx = 0.0f;
    constexpr int NPIX = GC_PH * GC_PW;
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    half_t *Xh = reinterpret_cast<half_t *>(gsm);      // [NPIX][GCP]
    half_t *Xc = Xh + NPIX * GCP;                      // corr units, same records
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swz = xcd_swizzle_c(blockIdx.x, gridDim.x);
    const int tx = swz % tiles_x, ty = swz / tiles_x;
    const int oy0 = row0 + ty * CTH, ox0 = tx * CTW;
    const int g = lane >> 4, lcol = lane & 15;

    constexpr int NLD = (NPIX * 8 + CNT - 1) / CNT;
    uint4 pre[NLD], prc[NLD];
#define GC_FETCH(chunk_)                                                                                  \
    _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                                     \
        const int p = tid + k * CNT;                                                                      \
        const int q = p >> 3, part = p & 7;                                                               \
        const int py = q / GC_PW, px = q - py * GC_PW;                                                    \
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;                                                   \
        uint4 v = make_uint4(0, 0, 0, 0), vc = make_uint4(0, 0, 0, 0);                                    \
        if (p < NPIX * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W) {                                     \
            const size_t o = (size_t)(iy * W + ix) * 256 + (chunk_)*64 + part * 8;                        \
            v = *reinterpret_cast<const uint4 *>(in + o);                                                 \
            if (IN_C) vc = *reinterpret_cast<const uint4 *>(in_c + o);                                    \
        }                                                                                                 \
        pre[k] = v; prc[k] = vc;                                                                          \
    }
    GC_FETCH(0)
    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int p = tid + k * CNT;
            if (p < NPIX * 8) {
                *reinterpret_cast<uint4 *>(Xh + (p >> 3) * GCP + (p & 7) * 8) = pre[k];
                if (IN_C) *reinterpret_cast<uint4 *>(Xc + (p >> 3) * GCP + (p & 7) * 8) = prc[k];
            }
        }
        const int pair = chunk * 4 + wave;
        h8_t wh[5];
        v8i_t wc8[3];
#pragma unroll
        for (int s = 0; s < 5; ++s)
            wh[s] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(pair * 5 + s) * 64 + lane) * 8);
        h8_t wl[5];
        if (IN_C && !X3) {
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const unsigned char *p = wck + ((size_t)(pair * 3 + m) * 64 + lane) * 32;
                wc8[m] = sfd2_cat8(*reinterpret_cast<const h8_t *>(p), *reinterpret_cast<const h8_t *>(p + 16));
            }
        } else {
#pragma unroll
            for (int s = 0; s < 5; ++s)
                wl[s] = *reinterpret_cast<const h8_t *>(reinterpret_cast<const half_t *>(wck) + ((size_t)(pair * 5 + s) * 64 + lane) * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (chunk + 1 < 4) { GC_FETCH(chunk + 1) }

        f32x4c_t acc[8], acl[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { acc[t] = (f32x4c_t){0.0f, 0.0f, 0.0f, 0.0f}; acl[t] = acc[t]; }
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            int tap = 2 * s + (g >> 1);
            if (tap > 8) tap = 8;  // zero-weight slot: read any valid location
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = ((t >> 1) + ky) * GC_PW + (t & 1) * 16 + lcol + kx;
                const h8_t bh = *reinterpret_cast<const h8_t *>(Xh + q * GCP + wave * 16 + (g & 1) * 8);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh, acc[t], 0, 0, 0);
                if (!IN_C || X3) acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh, acl[t], 0, 0, 0);
                if (X3) {
                    const h8_t bl = *reinterpret_cast<const h8_t *>(Xc + q * GCP + wave * 16 + (g & 1) * 8);
                    acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bl, acl[t], 0, 0, 0);
                }
            }
        }
        if (!IN_C || X3) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], 1.0f / 2048.0f, acc[t][r]);
        }
#pragma unroll
        for (int m = 0; m < (IN_C && !X3 ? 3 : 0); ++m) {
            int tap = 4 * m + g;
            if (tap > 8) tap = 8;
            const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int q = ((t >> 1) + ky) * GC_PW + (t & 1) * 16 + lcol + kx;
                const half_t *bp = Xc + q * GCP + wave * 16;
                const v8i_t bc = sfd2_cat8(*reinterpret_cast<const h8_t *>(bp), *reinterpret_cast<const h8_t *>(bp + 8));
                acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wc8[m], bc, acc[t], 0, 0, 0, sa, 0, 0x7f7f7f7f);
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(acc[t]));   // (scaled MFMAs are pure nodes: keep them in front of the barrier)
        const int c0 = pair * 16 + g * 4;
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
        // lane groups g and g^1 (16 lanes apart) hold adjacent 8-byte channel runs of the same pixel: exchange across two
        // pixel tiles so that every lane issues one 16-byte store per tile pair and plane
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
            uint2 pk[2], ck[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (X3) {
                    const float v0 = fmaxf(acc[t + j][0] * sc.x + sh.x, 0.0f), v1 = fmaxf(acc[t + j][1] * sc.y + sh.y, 0.0f);
                    const float v2 = fmaxf(acc[t + j][2] * sc.z + sh.z, 0.0f), v3 = fmaxf(acc[t + j][3] * sc.w + sh.w, 0.0f);
                    const h4_t h = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                    const h4_t l = {(half_t)((v0 - (float)h[0]) * 2048.0f), (half_t)((v1 - (float)h[1]) * 2048.0f),
                                    (half_t)((v2 - (float)h[2]) * 2048.0f), (half_t)((v3 - (float)h[3]) * 2048.0f)};
                    __builtin_memcpy(&pk[j], &h, 8);
                    __builtin_memcpy(&ck[j], &l, 8);
                } else {
                    sfd2_epi4<false>(acc[t + j][0], acc[t + j][1], acc[t + j][2], acc[t + j][3], sc, sh, sc, 0.0f, pk[j], ck[j], mx,
                                     oy0 + ((t + j) >> 1) < row1 && ox0 + ((t + j) & 1) * 16 + lcol < W);
                }
            }
            const bool odd = g & 1;
            const uint2 send = odd ? pk[0] : pk[1], sendc = odd ? ck[0] : ck[1];
            uint2 recv, recvc = make_uint2(0, 0);
            recv.x = __shfl_xor(send.x, 16); recv.y = __shfl_xor(send.y, 16);
            if (OUT_C) { recvc.x = __shfl_xor(sendc.x, 16); recvc.y = __shfl_xor(sendc.y, 16); }
            const int tt = odd ? t + 1 : t;                         // the tile this lane stores
            const int oy = oy0 + (tt >> 1), ox = ox0 + (tt & 1) * 16 + lcol;
            const uint4 v = odd ? make_uint4(recv.x, recv.y, pk[1].x, pk[1].y) : make_uint4(pk[0].x, pk[0].y, recv.x, recv.y);
            const uint4 vc = odd ? make_uint4(recvc.x, recvc.y, ck[1].x, ck[1].y) : make_uint4(ck[0].x, ck[0].y, recvc.x, recvc.y);
            if (oy < row1 && ox < W) {
                const size_t o = ((size_t)oy * W + ox) * 256 + pair * 16 + (g & ~1) * 4;
                *reinterpret_cast<uint4 *>(out + o) = v;
                if (OUT_C) *reinterpret_cast<uint4 *>(out_c + o) = vc;
            }
        }
    }
    if (!X3) sfd2_range_commit(range, sfd2_wave_max_bits(mx));
#undef GC_FETCH
}

// row0 / row1: the output rows to produce (the whole image: 0, H); input rows outside [0, H) are the conv's zero padding
// (sbyte < 0: SFD2_PREC_F16X3 -- in_c / out_c are lo' planes, wck the filters' lo' fragments)
void launch_gconv_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, const half_t *wpk, const void *wck,
                    const float *scale, const float *shift, half_t *out, half_t *out_c, int sbyte, int row0, int row1, unsigned int *range)
{
    // in_c == null: plain input, wck = fp16 residual fragments; out_c == null: hi plane only
    const size_t lds = (size_t)(in_c ? 2 : 1) * GC_PH * GC_PW * GCP * sizeof(half_t);
    if (row1 > H) row1 = H;
    if (row0 >= row1) return;
    const int tiles_x = (W + CTW - 1) / CTW, tiles_y = (row1 - row0 + CTH - 1) / CTH;
#define GCC_GO(...) hipLaunchKernelGGL((gconv_c_kernel<__VA_ARGS__>), dim3(tiles_x * tiles_y), dim3(CNT), lds, st, in, in_c, H, W, wpk, \
                       reinterpret_cast<const unsigned char *>(wck), scale, shift, out, out_c, tiles_x, (sbyte & 255) * 0x01010101, row0, row1, range)
    if (sbyte < 0) {
        if (!in_c || !out_c) abort();
        GCC_GO(true, true, true);
    } else if (in_c && out_c) GCC_GO(true, true);
    else if (in_c) GCC_GO(true, false);
    else if (!out_c) GCC_GO(false, false);
    else abort();
#undef GCC_GO
}

// hi + corr planes -> NCHW fp32 (sfd2_debug_activation): hi + the residual the corr unit carries
// fmt6: the corr plane holds fp6 half-records (sfd2_epi16_fp6): per 32-channel chunk 64 bytes, half-record h in the 16-byte slots h and 2 + h --
// 32 six-bit codes (2 j = residual * 2^11, 2 j + 1 = value; j = 4 q + r <-> channel 8 q + 4 h + r), then the block's E8M0 scale byte
__global__ void nhwc_hc_to_nchw_f_kernel(const half_t *__restrict__ in, const half_t *__restrict__ in_c, int npix, int pitch, int c,
                                         float *__restrict__ out, int fmt6)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)npix * c) return;
    const int ch = (int)(i / npix), p = (int)(i % npix);
    const size_t o = (size_t)p * pitch + ch;
    if (fmt6 == 2) {      // one residual byte per channel, c bytes per pixel (option "trunk_r1")
        const unsigned char rb = reinterpret_cast<const unsigned char *>(in_c)[(size_t)p * pitch + ch];
        out[i] = (float)in[o] + __builtin_amdgcn_cvt_f32_fp8((int)rb, 0) * (1.0f / (float)(1 << SFD2_C_XL_SHIFT));
        return;
    }
    if (fmt6) {
        const int cl = ch & 31, q = cl >> 3, h = (cl >> 2) & 1, r = cl & 3, j = 4 * q + r;
        const unsigned char *rec = reinterpret_cast<const unsigned char *>(in_c) + ((size_t)p * pitch + (ch & ~31)) * 2;
        const int bit = 12 * j;                                   // code 2 j of the 192-bit string: bytes 0..15 in slot h, 16..23 in slot 2 + h
        auto byte_at = [&](int b) -> unsigned { return b < 16 ? rec[16 * h + b] : rec[32 + 16 * h + (b - 16)]; };
        const unsigned w16 = byte_at(bit >> 3) | (byte_at((bit >> 3) + 1) << 8);
        const unsigned code = (w16 >> (bit & 7)) & 63u;
        const unsigned e8 = rec[32 + 16 * h + 8];
        const int ex = (int)((code >> 3) & 3u), mant = (int)(code & 7u);
#if SFD2_PIX6_BF6
        const int ex3 = (int)((code >> 2) & 7u), m2 = (int)(code & 3u);
        float v = ex3 ? (float)(4 + m2) * 0.25f * __builtin_ldexpf(1.0f, ex3 - 3) : (float)m2 * 0.0625f;
        (void)ex; (void)mant;
#else
        float v = ex ? (float)(8 + mant) * 0.125f * (float)(1 << (ex - 1)) : (float)mant * 0.125f;
#endif
        if (code & 32u) v = -v;
        out[i] = (float)in[o] + v * __builtin_ldexpf(1.0f, (int)e8 - 127 - 11);
        return;
    }
    const unsigned short u = reinterpret_cast<const unsigned short *>(in_c)[o];
    out[i] = (float)in[o] + sfd2_corr_lo((unsigned)u, 0);
}
void launch_nhwc_hc_to_nchw_f(hipStream_t st, const half_t *in, const half_t *in_c, int npix, int pitch, int c, float *out, int fmt6)
{
    const size_t n = (size_t)npix * c;
    hipLaunchKernelGGL(nhwc_hc_to_nchw_f_kernel, dim3((unsigned)((n + CNT - 1) / CNT)), dim3(CNT), 0, st, in, in_c, npix, pitch, c, out, fmt6);
}
