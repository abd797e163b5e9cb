// Small helpers of libsfd2hip on gfx950 that are not on the per-image hot path: layout conversions of the parity / debug entry
// points, the absolute maximum of an activation tensor (range calibration of the compensated mode), and the decoder-side ingest
// (uint8 HWC -> float32 CHW with OpenCV's cubic resize_max).
#include "sfd2_internal.h"
#include <math.h>
#include <stdlib.h>

#define NT 256

// ---------------------------------------------------------------- layout helpers (parity / debug paths)
__global__ void nhwc_h_to_nchw_f_kernel(const half_t *__restrict__ in, int npix, int pitch, int c, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)npix * c) return;
    const int ch = (int)(i / npix), p = (int)(i % npix);
    out[i] = (float)in[(size_t)p * pitch + ch];
}
__global__ void nhwc_f_to_nchw_f_kernel(const float *__restrict__ in, int npix, int pitch, int c, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)npix * c) return;
    const int ch = (int)(i / npix), p = (int)(i % npix);
    out[i] = in[(size_t)p * pitch + ch];
}
__global__ void nchw_f_to_nhwc_f_kernel(const float *__restrict__ in, int npix, int c, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)npix * c) return;
    const int p = (int)(i / c), ch = (int)(i % c);
    out[i] = in[(size_t)ch * npix + p];
}
void launch_nhwc_h_to_nchw_f(hipStream_t st, const half_t *in, int npix, int pitch, int c, float *out)
{
    const size_t n = (size_t)npix * c;
    hipLaunchKernelGGL(nhwc_h_to_nchw_f_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, in, npix, pitch, c, out);
}
void launch_nhwc_f_to_nchw_f(hipStream_t st, const float *in, int npix, int pitch, int c, float *out)
{
    const size_t n = (size_t)npix * c;
    hipLaunchKernelGGL(nhwc_f_to_nchw_f_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, in, npix, pitch, c, out);
}
void launch_nchw_f_to_nhwc_f(hipStream_t st, const float *in, int npix, int c, float *out)
{
    const size_t n = (size_t)npix * c;
    hipLaunchKernelGGL(nchw_f_to_nhwc_f_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, in, npix, c, out);
}

// out[i] *= mul (sfd2_debug_activation: a tensor of the fp16 family is stored times 2^act_exp, the caller gets the network's values)
__global__ void scale_inplace_kernel(float *__restrict__ p, size_t n, float mul)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= mul;
}
void launch_scale_inplace(hipStream_t st, float *p, size_t n, float mul)
{
    if (n == 0 || mul == 1.0f) return;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, st, p, n, mul);
}

// ---------------------------------------------------------------- range calibration
// *out = max(*out, max |in[i]|) as the bit pattern of the non-negative float (unsigned order = float order); NaNs are skipped.
__global__ __launch_bounds__(NT)
void absmax_f32_kernel(const float *__restrict__ in, size_t n, unsigned int *__restrict__ out)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const float v = fabsf(in[i]);
        m = v > m ? v : m;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float o = __shfl_xor(m, d);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(out, __float_as_uint(m));
}
void launch_absmax_f32(hipStream_t st, const float *in, size_t n, unsigned int *out)
{
    if (n == 0) return;
    const size_t blocks = (n + NT * 8 - 1) / (NT * 8);
    hipLaunchKernelGGL(absmax_f32_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(NT), 0, st, in, n, out);
}

// ---------------------------------------------------------------- decoder-side ingest (extract_localization.py:158-186)
// ImageDataset.__getitem__: image.astype(float32) -> cv2.resize(..., INTER_CUBIC) when max(w, h) > resize_max ->
// HWC to CHW -> / 255.  OpenCV's float32 cubic resize, restated from its published algorithm (imgproc resize.cpp,
// resizeGeneric_ with HResizeCubic / VResizeCubic): separable, horizontal pass first, Keys kernel with A = -0.75,
// source coordinate (d + 0.5) * (src / dst) - 0.5 evaluated in double and rounded to float, taps sx-1 .. sx+2 with
// replicated borders, no clamping of the overshoot.  Products and sums in the order the scalar code writes them, with
// contraction forbidden (__fmul_rn / __fadd_rn) so the oracle's numpy restatement is reproduced bit for bit.
// HIP's __fmul_rn / __fadd_rn are header inlines compiled under the default contract mode, so they still fuse; the
// arithmetic below is therefore written with plain operators lexically under `fp contract(off)`.
__device__ __forceinline__ void cubic_coeffs(float x, float *c)
{
#pragma clang fp contract(off)
    const float A = -0.75f;
    const float x1 = x + 1.0f;
    c[0] = ((A * x1 - 5.0f * A) * x1 + 8.0f * A) * x1 - 4.0f * A;
    c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    const float y = 1.0f - x;
    c[2] = ((A + 2.0f) * y - (A + 3.0f)) * y * y + 1.0f;
    c[3] = 1.0f - c[0] - c[1] - c[2];
}

__global__ __launch_bounds__(NT)
void ingest_u8_kernel(const unsigned char *__restrict__ src, int H, int W, int bgr, int nh, int nw, double scale_x,
                      double scale_y, float *__restrict__ out /*[3][nh][nw]*/)
{
#pragma clang fp contract(off)
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (ox >= nw) return;
    const size_t plane = (size_t)nh * nw;
    if (nh == H && nw == W) {   // no resize: astype(float32) / 255.
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)src[((size_t)oy * W + ox) * 3 + (bgr ? 2 - c : c)];
            out[c * plane + (size_t)oy * nw + ox] = __fdiv_rn(v, 255.0f);
        }
        return;
    }
    const double dx = (ox + 0.5) * scale_x, dy = (oy + 0.5) * scale_y;
    float fx = (float)(dx - 0.5);
    float fy = (float)(dy - 0.5);
    const int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx = fx - (float)sx;
    fy = fy - (float)sy;
    float a[4], b[4];
    cubic_coeffs(fx, a);
    cubic_coeffs(fy, b);
    int xs[4], ys[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xs[k] = min(max(sx - 1 + k, 0), W - 1);
        ys[k] = min(max(sy - 1 + k, 0), H - 1);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cs = bgr ? 2 - c : c;
        float rows[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned char *row = src + (size_t)ys[r] * W * 3 + cs;
            float acc = (float)row[xs[0] * 3] * a[0];
            acc = acc + (float)row[xs[1] * 3] * a[1];
            acc = acc + (float)row[xs[2] * 3] * a[2];
            acc = acc + (float)row[xs[3] * 3] * a[3];
            rows[r] = acc;
        }
        float v = rows[0] * b[0];
        v = v + rows[1] * b[1];
        v = v + rows[2] * b[2];
        v = v + rows[3] * b[3];
        out[c * plane + (size_t)oy * nw + ox] = __fdiv_rn(v, 255.0f);
    }
}

// SFD2_FLAG_IMG_U8_X: [n][4] bytes -> [n][3] bytes (the fourth byte of a pixel dropped); four pixels per thread
__global__ __launch_bounds__(256)
void unpack_rgbx_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst, size_t npix)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;          // pixel quad
    const size_t p0 = q * 4;
    if (p0 + 4 <= npix && ((reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) & 15) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + p0 * 4);
        // bytes r0 g0 b0 x | r1 g1 b1 x | r2 g2 b2 x | r3 g3 b3 x  ->  r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const unsigned int a = (v.x & 0x00FFFFFFu) | (v.y << 24);
        const unsigned int b = ((v.y >> 8) & 0x0000FFFFu) | (v.z << 16);
        const unsigned int c = ((v.z >> 16) & 0x000000FFu) | (v.w << 8);
        unsigned int *d = reinterpret_cast<unsigned int *>(dst + p0 * 3);
        d[0] = a; d[1] = b; d[2] = c;
    } else {
        for (size_t p = p0; p < npix && p < p0 + 4; ++p)
            for (int ch = 0; ch < 3; ++ch) dst[p * 3 + ch] = src[p * 4 + ch];
    }
}
void launch_unpack_rgbx(hipStream_t st, const unsigned char *src, unsigned char *dst, size_t npix)
{
    const size_t quads = (npix + 3) / 4;
    hipLaunchKernelGGL(unpack_rgbx_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, src, dst, npix);
}

void launch_ingest_u8(hipStream_t st, const unsigned char *src, int H, int W, int bgr, int nh, int nw, float *out)
{
    // cv2.resize: inv_scale = dsize / ssize (double), scale = 1. / inv_scale
    const double scale_x = 1.0 / ((double)nw / (double)W), scale_y = 1.0 / ((double)nh / (double)H);
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((nw + NT - 1) / NT, nh), dim3(NT), 0, st, src, H, W, bgr, nh, nw, scale_x, scale_y, out);
}

// ---------------------------------------------------------------- per-image record of an asynchronous extract
// One wave.  Lane t < SFD2_RS_COUNT folds tensor t's SFD2_RANGE_SUB running maxima (stored units, bit patterns of non-negative
// floats: unsigned order = float order, and every Inf / NaN pattern compares above the saturation value) into the device-side
// history hist[t] and -- for a tensor that reached SFD2_C_SAT -- CLEARS them, so that the next image's record of that tensor is that image's alone.  A tensor
// below the saturation keeps its running maxima: they are what lets sfd2_range_commit skip its atomic (a wave only writes a value ABOVE what is recorded).
// Cleared after every image (rounds 4 - 5) every wave of every recording kernel issued its atomicMax again, image after image: conv1x1_c256_c 42 -> 64 us,
// the stem 146 -> 170, rb23 73 -> 81, 0.2 ms per 1600x1200 extract in the pipelined driver and in every synchronous extract (profiles/r05l_io_modes2.txt).
// The per-image saturation mask needs no more than this: below the saturation the running maximum says "no image so far", at or above it the image that
// got it there is the one being recorded, and its words are cleared.  rec (null = fold only) receives
// { key points (clipped to sel_cap), NMS survivors, mask of the tensors that reached SFD2_C_SAT, bit 0: candidate overflow }.
__global__ __launch_bounds__(64)
void extract_record_kernel(unsigned int *__restrict__ range_stat, unsigned int *__restrict__ hist, const unsigned int *__restrict__ counters,
                           int sel_cap, int cand_cap, unsigned int *__restrict__ rec)
{
    const int t = threadIdx.x;
    unsigned int m = 0;
    if (t < SFD2_RS_COUNT) {
#pragma unroll
        for (int s = 0; s < SFD2_RANGE_SUB; ++s) m = max(m, range_stat[t * SFD2_RANGE_SUB + s]);
        if (m >= __float_as_uint(SFD2_C_SAT)) {
#pragma unroll
            for (int s = 0; s < SFD2_RANGE_SUB; ++s) range_stat[t * SFD2_RANGE_SUB + s] = 0u;
        }
        hist[t] = max(hist[t], m);
    }
    const unsigned long long sat = __ballot(t < SFD2_RS_COUNT && m >= __float_as_uint(SFD2_C_SAT));
    if (t == 0 && rec) {
        const unsigned int n = counters ? counters[1] : 0u, nc = counters ? counters[0] : 0u;
        rec[0] = min(n, (unsigned int)sel_cap);
        rec[1] = nc;
        rec[2] = (unsigned int)sat;
        rec[3] = nc > (unsigned int)cand_cap ? 1u : 0u;
    }
}
void launch_extract_record(hipStream_t st, unsigned int *range_stat, unsigned int *hist, const unsigned int *counters, int sel_cap,
                           int cand_cap, unsigned int *rec)
{
    hipLaunchKernelGGL(extract_record_kernel, dim3(1), dim3(64), 0, st, range_stat, hist, counters, sel_cap, cand_cap, rec);
}
