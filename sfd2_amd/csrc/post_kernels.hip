// Heads and key-point selection of the SFD2 extractor on gfx950: detector soft-max +
// depth-to-space, stability up-sampling + arg-max, heat map, max-pool NMS, threshold /
// border / top-K selection with a defined tie order, bilinear descriptor sampling.
// HBM-bound work: coalesced loads, LDS tiling, wave64 reductions.
//
// Replaces (file:line in the reference):
//   nets/sfd2.py:329-337 (detector head)      nets/sfd2.py:305-311,345-347 (stability)
//   nets/extractor.py:137-141 (heat map)       nets/extractor.py:20-35 (simple_nms)
//   nets/extractor.py:158-183,322-326 (selection)   nets/extractor.py:199-208 (descriptors)
#include "sfd2_internal.h"
#include <math.h>
#include <stdlib.h>

#define NT 256

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ---------------------------------------------------------------- detector head
// One wave per coarse cell: lane c holds channel c (0..63), channel 64 is the dust bin.
__global__ __launch_bounds__(NT)
void detector_head_kernel(const float *__restrict__ logits, int pitch, int hc, int wc, float *__restrict__ score)
{
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (cell >= hc * wc) return;
    const int y = cell / wc, x = cell - y * wc;
    const float *p = logits + (size_t)cell * pitch;
    const float e = expf(p[lane]);
    const float ed = expf(p[64]);
    const float den = (wave_sum(e) + ed) + 0.00001f;
    score[(size_t)(8 * y + (lane >> 3)) * (8 * wc) + 8 * x + (lane & 7)] = e / den;
}

void launch_detector_head(hipStream_t st, const float *logits, int pitch, int hc, int wc, float *score)
{
    const int cells = hc * wc;
    hipLaunchKernelGGL(detector_head_kernel, dim3((cells + 3) / 4), dim3(NT), 0, st, logits, pitch, hc, wc, score);
}

// ---------------------------------------------------------------- bilinear (torch, align_corners=False)
// Rounding sequence pinned to torch's CPU kernel (see oracle/orc_post.c orc_resize_bilinear):
//   src = fma(scale, dst + 0.5, -0.5) clamped at 0;   out = fma(top, ly0, bot*ly1), top = fma(v00, lx0, v01*lx1)
struct LinCoef { int i0, i1; float l0, l1; };
__device__ __forceinline__ LinCoef lin_coef(int dst, int in_size, int out_size, float scale)
{
    LinCoef c;
    if (in_size == out_size) { c.i0 = dst; c.i1 = dst; c.l0 = 1.0f; c.l1 = 0.0f; return c; }
    float src = __fmaf_rn(scale, (float)dst + 0.5f, -0.5f);
    if (src < 0.0f) src = 0.0f;
    int a = (int)floorf(src);
    if (a > in_size - 1) a = in_size - 1;
    float lam = __fsub_rn(src, (float)a);
    lam = fminf(fmaxf(lam, 0.0f), 1.0f);
    c.i0 = a;
    c.i1 = (a + 1 < in_size) ? a + 1 : in_size - 1;
    c.l1 = lam;
    c.l0 = __fsub_rn(1.0f, lam);
    return c;
}
__device__ __forceinline__ float bilerp(const float *__restrict__ p, int w, const LinCoef &cy, const LinCoef &cx)
{
    const float v00 = p[(size_t)cy.i0 * w + cx.i0], v01 = p[(size_t)cy.i0 * w + cx.i1];
    const float v10 = p[(size_t)cy.i1 * w + cx.i0], v11 = p[(size_t)cy.i1 * w + cx.i1];
    const float top = __fmaf_rn(v00, cx.l0, __fmul_rn(v01, cx.l1));
    const float bot = __fmaf_rn(v10, cx.l0, __fmul_rn(v11, cx.l1));
    return __fmaf_rn(top, cy.l0, __fmul_rn(bot, cy.l1));
}

// heat[y][x] = resize(score)[y][x] * cls_to_value(argmax_c resize(sta_c)[y][x])
__global__ __launch_bounds__(NT)
void heatmap_kernel(const float *__restrict__ score, int hs, int ws, float sc_y, float sc_x,
                    const float *__restrict__ sta, int hc, int wc, float st_y, float st_x,
                    int H, int W, float *__restrict__ heat, float *__restrict__ stab_out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    float s;
    if (hs == H && ws == W) {
        s = score[(size_t)y * ws + x];
    } else {
        const LinCoef cy = lin_coef(y, hs, H, sc_y), cx = lin_coef(x, ws, W, sc_x);
        s = bilerp(score, ws, cy, cx);
    }
    float stab = 1.0f;
    if (sta) {
        const LinCoef cy = lin_coef(y, hc, H, st_y), cx = lin_coef(x, wc, W, st_x);
        const size_t plane = (size_t)hc * wc;
        const float v0 = bilerp(sta, wc, cy, cx);
        const float v1 = bilerp(sta + plane, wc, cy, cx);
        const float v2 = bilerp(sta + 2 * plane, wc, cy, cx);
        int best = 0;
        float bv = v0;
        if (v1 > bv) { bv = v1; best = 1; }
        if (v2 > bv) { bv = v2; best = 2; }
        stab = best == 0 ? 0.1f : (best == 1 ? 0.5f : 1.0f);
        if (stab_out) stab_out[(size_t)y * W + x] = stab;
    }
    if (heat) heat[(size_t)y * W + x] = __fmul_rn(s, stab);
}

void launch_heatmap(hipStream_t st, const float *score, int hs, int ws, const float *sta, int hc, int wc,
                    int H, int W, float *heat, float *stab_out)
{
    const float sc_y = (float)hs / (float)H, sc_x = (float)ws / (float)W;
    const float st_y = hc > 0 ? (float)hc / (float)H : 1.0f, st_x = wc > 0 ? (float)wc / (float)W : 1.0f;
    hipLaunchKernelGGL(heatmap_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(NT), 0, st, score, hs, ws, sc_y, sc_x,
                       sta, hc, wc, st_y, st_x, H, W, heat, stab_out);
}

// detector head + heat map in one pass (extract path, H == 8 * hc8, W == 8 * wc8 so the score map needs no resize):
// one wave per 8 x 8 cell as detector_head_kernel, the lane then weights its pixel with the stability value exactly as
// heatmap_kernel does -- same operations, same results, and the 7.7 MB score map is neither written nor read back.
// (Folding this into the NMS tile load as well was measured and lost: the NMS regions overlap 3.9-fold, so the per-cell
// soft-max would be evaluated 4.8 times over: 69 -> 76 us for the three stages.)
__global__ __launch_bounds__(NT)
void heads_heat_kernel(const float *__restrict__ logits, int pitch, int hc8, int wc8, const float *__restrict__ sta, int hc, int wc,
                       float st_y, float st_x, int H, int W, float *__restrict__ heat)
{
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (cell >= hc8 * wc8) return;
    const int cy = cell / wc8, cx = cell - cy * wc8;
    const float *p = logits + (size_t)cell * pitch;
    const float e = expf(p[lane]);
    const float ed = expf(p[64]);
    const float den = (wave_sum(e) + ed) + 0.00001f;
    const float s = e / den;
    const int y = 8 * cy + (lane >> 3), x = 8 * cx + (lane & 7);
    float stab = 1.0f;
    if (sta) {
        const LinCoef ly = lin_coef(y, hc, H, st_y), lx = lin_coef(x, wc, W, st_x);
        const size_t plane = (size_t)hc * wc;
        const float v0 = bilerp(sta, wc, ly, lx);
        const float v1 = bilerp(sta + plane, wc, ly, lx);
        const float v2 = bilerp(sta + 2 * plane, wc, ly, lx);
        int best = 0;
        float bv = v0;
        if (v1 > bv) { bv = v1; best = 1; }
        if (v2 > bv) { bv = v2; best = 2; }
        stab = best == 0 ? 0.1f : (best == 1 ? 0.5f : 1.0f);
    }
    heat[(size_t)y * W + x] = __fmul_rn(s, stab);
}

void launch_heads_heat(hipStream_t st, const float *logits, int pitch, int hc8, int wc8, const float *sta, int hc, int wc,
                       int H, int W, float *heat)
{
    const float st_y = hc > 0 ? (float)hc / (float)H : 1.0f, st_x = wc > 0 ? (float)wc / (float)W : 1.0f;   // as launch_heatmap
    const int cells = hc8 * wc8;
    hipLaunchKernelGGL(heads_heat_kernel, dim3((cells + 3) / 4), dim3(NT), 0, st, logits, pitch, hc8, wc8, sta, hc, wc, st_y, st_x, H, W, heat);
}

// convPb + detector head + heat map in one kernel (extract path, H and W multiples of 8).  A block takes 32 consecutive
// 8 x 8 cells: their 256-channel convPa.3 vectors go to LDS, waves 0-2 run convPb on them with MFMAs (65 real of 96
// computed output channels x 32 cells x K = 256; filter fragments straight from the packed filters in L2), the fp32
// logits are parked in LDS, and every wave then finishes eight cells exactly as heads_heat_kernel does.  K ascends in
// 16-wide slices into one accumulator and the epilogue is acc * scale + shift, as conv_igemm2 computes the logits, so the
// heat map is bit-identical; the 15 MB logit tensor (128-channel pitch for 65 logits) is neither written nor read.
#define PH_CELLS 32
#define PH_XREC 528      // bytes per gathered cell record: 512 + 16 pad
#define PH_OREC 100      // floats per logit record: 96 + 4 pad
__global__ __launch_bounds__(NT)
void pb_heads_heat_kernel(const half_t *__restrict__ fmap /*[cells][256]*/, int hc8, int wc8,
                          const half_t *__restrict__ wpk /*[8 chunks][CoutP][32]*/, int CoutP, const float *__restrict__ scale,
                          const float *__restrict__ shift, const float *__restrict__ sta, int hc, int wc, float st_y, float st_x,
                          int H, int W, float *__restrict__ heat, unsigned int *__restrict__ zero_words, int n_zero)
{
    // the selection counters + score histogram that the NMS kernel behind this one appends to (saves a memset node)
    if (zero_words && (int)(blockIdx.x * blockDim.x + threadIdx.x) < n_zero) zero_words[blockIdx.x * blockDim.x + threadIdx.x] = 0u;
    __shared__ __attribute__((aligned(16))) unsigned char X[PH_CELLS * PH_XREC];
    __shared__ __attribute__((aligned(16))) float O[PH_CELLS * PH_OREC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lhi = lane >> 5;
    const int ncell = hc8 * wc8, c0cell = blockIdx.x * PH_CELLS;

    h8_t a[16];
    if (wave < 3) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            a[kk] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(kk >> 1) * CoutP + wave * 32 + lrow) * 32 + (kk & 1) * 16 + lhi * 8);
    }
    for (int rec = wave * 2 + lhi; rec < PH_CELLS; rec += 8) {
        const int cell = c0cell + rec;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (cell < ncell) v = *reinterpret_cast<const uint4 *>(fmap + (size_t)cell * 256 + lrow * 8);
        *reinterpret_cast<uint4 *>(X + rec * PH_XREC + lrow * 16) = v;
    }
    __syncthreads();
    if (wave < 3) {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const h8_t b = *reinterpret_cast<const h8_t *>(X + lrow * PH_XREC + kk * 32 + lhi * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = wave * 32 + 8 * q + 4 * lhi;
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
            const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
            *reinterpret_cast<float4 *>(O + lrow * PH_OREC + c0) =
                make_float4(acc[4 * q + 0] * sc.x + sh.x, acc[4 * q + 1] * sc.y + sh.y, acc[4 * q + 2] * sc.z + sh.z, acc[4 * q + 3] * sc.w + sh.w);
        }
    }
    __syncthreads();
    // (branch-free and unrolled by four: the twelve stability-map loads of a cell are then in flight for four cells at a
    // time instead of one memory round trip per cell)
#pragma unroll 4
    for (int jj = 0; jj < PH_CELLS / (NT / 64); ++jj) {
        const int j = wave + jj * (NT / 64);
        const bool live = c0cell + j < ncell;
        const int cell = live ? c0cell + j : ncell - 1;
        const int cy = cell / wc8, cx = cell - cy * wc8;
        const float *p = O + j * PH_OREC;
        const float e = expf(p[lane]);
        const float ed = expf(p[64]);
        const float den = (wave_sum(e) + ed) + 0.00001f;
        const float s = e / den;
        const int y = 8 * cy + (lane >> 3), x = 8 * cx + (lane & 7);
        float stab = 1.0f;
        if (sta) {
            const LinCoef ly = lin_coef(y, hc, H, st_y), lx = lin_coef(x, wc, W, st_x);
            const size_t plane = (size_t)hc * wc;
            const float v0 = bilerp(sta, wc, ly, lx);
            const float v1 = bilerp(sta + plane, wc, ly, lx);
            const float v2 = bilerp(sta + 2 * plane, wc, ly, lx);
            int best = 0;
            float bv = v0;
            if (v1 > bv) { bv = v1; best = 1; }
            if (v2 > bv) { bv = v2; best = 2; }
            stab = best == 0 ? 0.1f : (best == 1 ? 0.5f : 1.0f);
        }
        if (live) heat[(size_t)y * W + x] = __fmul_rn(s, stab);
    }
}

void launch_pb_heads_heat(hipStream_t st, const half_t *fmap, int hc8, int wc8, const half_t *wpk, int CoutP, const float *scale,
                          const float *shift, const float *sta, int hc, int wc, int H, int W, float *heat, unsigned int *zero_words,
                          int n_zero)
{
    const float st_y = hc > 0 ? (float)hc / (float)H : 1.0f, st_x = wc > 0 ? (float)wc / (float)W : 1.0f;   // as launch_heatmap
    const int cells = hc8 * wc8;
    const int grid = (cells + PH_CELLS - 1) / PH_CELLS;
    if (grid * NT < n_zero) zero_words = nullptr;          // (tiny images: the caller keeps its memset)
    hipLaunchKernelGGL(pb_heads_heat_kernel, dim3(grid), dim3(NT), 0, st, fmap, hc8, wc8, wpk, CoutP,
                       scale, shift, sta, hc, wc, st_y, st_x, H, W, heat, zero_words, n_zero);
}

bool pb_heads_heat_clears(int hc8, int wc8, int n_zero) { return ((hc8 * wc8 + PH_CELLS - 1) / PH_CELLS) * NT >= n_zero; }

// ---------------------------------------------------------------- simple_nms (+ threshold/border/compaction)
// One block = 32 x 64 output pixels, LDS region = tile + 5*radius halo (radius 4 -> 72 x 104):
// five chained 9x9 max-pools, each run as a row pass and a column pass over the whole region; a
// value is exact once it is 4 px further from the region edge than its inputs.
// Padding semantics of torch.max_pool2d (-inf outside the image) are carried by the data:
// scores outside the image are -inf, masks outside the image are 0.
#define NMS_TH 32
#define NMS_TW 64
#define NMS_RMAX 4
#define NMS_RH (NMS_TH + 10 * NMS_RMAX)
#define NMS_RW (NMS_TW + 10 * NMS_RMAX)
#define NMS_N (NMS_RH * NMS_RW)

__device__ __forceinline__ void pool_rows(const float *__restrict__ src, float *__restrict__ dst, int r)
{
    for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
        const int y = i / NMS_RW, x = i - y * NMS_RW;
        const int a = x - r < 0 ? 0 : x - r, b = x + r >= NMS_RW ? NMS_RW - 1 : x + r;
        float m = src[y * NMS_RW + a];
        for (int t = a + 1; t <= b; ++t) m = fmaxf(m, src[y * NMS_RW + t]);
        dst[i] = m;
    }
}
__device__ __forceinline__ void pool_cols(const float *__restrict__ src, float *__restrict__ dst, int r)
{
    for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
        const int y = i / NMS_RW, x = i - y * NMS_RW;
        const int a = y - r < 0 ? 0 : y - r, b = y + r >= NMS_RH ? NMS_RH - 1 : y + r;
        float m = src[a * NMS_RW + x];
        for (int t = a + 1; t <= b; ++t) m = fmaxf(m, src[t * NMS_RW + x]);
        dst[i] = m;
    }
}

__global__ __launch_bounds__(512)
void nms_select_kernel(const float *__restrict__ heat, int H, int W, int radius, float conf_th, int border, int Hb, int Wb,
                       float *__restrict__ nms_dense, unsigned long long *__restrict__ cand, int cand_cap,
                       unsigned int *__restrict__ counters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *S = reinterpret_cast<float *>(smem);
    float *A = S + NMS_N;
    float *B = A + NMS_N;
    unsigned char *M = reinterpret_cast<unsigned char *>(B + NMS_N);
    const int halo = 5 * radius;
    const int gy0 = blockIdx.y * NMS_TH - halo, gx0 = blockIdx.x * NMS_TW - halo;
    const float NEG = -INFINITY;

    for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
        const int y = i / NMS_RW, x = i - y * NMS_RW;
        const int gy = gy0 + y, gx = gx0 + x;
        S[i] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? heat[(size_t)gy * W + gx] : NEG;
    }
    __syncthreads();
    pool_rows(S, A, radius);
    __syncthreads();
    pool_cols(A, B, radius);
    __syncthreads();
    for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
        const bool oob = S[i] == NEG;
        const bool mk = !oob && (S[i] == B[i]);            // max_mask = scores == max_pool(scores)
        M[i] = mk ? 1 : 0;
        B[i] = mk ? 1.0f : 0.0f;
    }
    __syncthreads();
    for (int it = 0; it < 2; ++it) {
        pool_rows(B, A, radius);
        __syncthreads();
        pool_cols(A, B, radius);                           // B = max_pool(max_mask.float())
        __syncthreads();
        for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
            const bool oob = S[i] == NEG;
            const bool supp = B[i] > 0.0f;                 // supp_mask
            M[i] = (M[i] & 1) | (supp ? 2 : 0);
            B[i] = oob ? NEG : (supp ? 0.0f : S[i]);       // supp_scores (padding stays -inf)
        }
        __syncthreads();
        pool_rows(B, A, radius);
        __syncthreads();
        pool_cols(A, B, radius);                           // B = max_pool(supp_scores)
        __syncthreads();
        for (int i = threadIdx.x; i < NMS_N; i += blockDim.x) {
            const bool oob = S[i] == NEG;
            const bool supp = (M[i] & 2) != 0;
            const float ss = supp ? 0.0f : S[i];
            const bool new_max = (ss == B[i]);
            bool mk = (M[i] & 1) != 0;
            if (!oob && new_max && !supp) mk = true;       // max_mask | (new_max_mask & ~supp_mask)
            M[i] = mk ? 1 : 0;
            B[i] = mk ? 1.0f : 0.0f;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < NMS_TH * NMS_TW; i += blockDim.x) {
        const int ty = i / NMS_TW, tx = i - ty * NMS_TW;
        const int gy = blockIdx.y * NMS_TH + ty, gx = blockIdx.x * NMS_TW + tx;
        if (gy >= H || gx >= W) continue;
        const int li = (ty + halo) * NMS_RW + tx + halo;
        const float v = M[li] ? S[li] : 0.0f;              // where(max_mask, scores, zeros)
        if (nms_dense) nms_dense[(size_t)gy * W + gx] = v;
        if (cand && v > conf_th && gx >= border && gx < Wb - border && gy >= border && gy < Hb - border) {
            const unsigned int idx = (unsigned int)(gy * W + gx);
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
            const unsigned int pos = atomicAdd(&counters[0], 1u);
            if (pos < (unsigned int)cand_cap) cand[pos] = key;
        }
    }
}

// generic-radius kernel also feeds the histogram (same key -> bin map)
__global__ __launch_bounds__(NT)
void hist_from_cand_kernel(const unsigned long long *__restrict__ cand, int cand_cap,
                           const unsigned int *__restrict__ counters, unsigned int *__restrict__ hist)
{
    unsigned int n = counters[0];
    if (n > (unsigned int)cand_cap) n = cand_cap;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&hist[key_bin(cand[i])], 1u);
}

bool launch_nms_select(hipStream_t st, const float *heat, int H, int W, int radius, float conf_th, int border, int Hb, int Wb,
                       float *nms_dense, unsigned long long *cand, int cand_cap, unsigned int *counters, int fuse_threshold, int top_k)
{
    unsigned int *hist = counters + 16;   // counters[0..15], then SFD2_HIST_BINS histogram bins
    if (radius == 4) {   // nms4_kernels.hip
        launch_nms4_select(st, heat, H, W, conf_th, border, Hb, Wb, nms_dense, cand, cand_cap, counters, hist, fuse_threshold, top_k);
        return fuse_threshold != 0 && cand != nullptr;
    }
    static bool attr_done = false;
    const size_t lds = (size_t)NMS_N * (3 * sizeof(float) + 1);
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(nms_select_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL(nms_select_kernel, dim3((W + NMS_TW - 1) / NMS_TW, (H + NMS_TH - 1) / NMS_TH), dim3(512), lds,
                       st, heat, H, W, radius, conf_th, border, Hb, Wb, nms_dense, cand, cand_cap, counters);
    if (cand) hipLaunchKernelGGL(hist_from_cand_kernel, dim3(64), dim3(NT), 0, st, cand, cand_cap, counters, hist);
    return false;
}

// ---------------------------------------------------------------- top-K + sort
// Keys are unique 64-bit integers: (score bits << 32) | (0xFFFFFFFF - pixel index); descending
// key order == score descending, then pixel index ascending (the tie rule of DESIGN.md).
// counters: [0] n_cand (may exceed cap: overflow flag), [1] n_selected (K), [2] cursor of keys above
//           the boundary bin, [3] boundary-list cursor, [4] boundary bin b*, [5] keys still needed
//           from bin b*, [6] select-all flag; counters[16 ..] = SFD2_HIST_BINS-bin histogram (key_bin)
//           (filled by the NMS kernel while it appends candidates).
__global__ __launch_bounds__(1024)
void select_threshold_kernel(int cand_cap, int top_k, unsigned int *__restrict__ counters)
{
    __shared__ unsigned int wsum[16];
    sfd2_select_threshold<false>(cand_cap, top_k, counters, wsum);
}

__global__ __launch_bounds__(NT)
void compact_selected_kernel(const unsigned long long *__restrict__ cand, int cand_cap,
                             unsigned long long *__restrict__ sel, int sel_cap,
                             unsigned long long *__restrict__ bnd, unsigned int *__restrict__ counters)
{
    unsigned int n = counters[0];
    if (n > (unsigned int)cand_cap) n = cand_cap;
    const unsigned int bstar = counters[4], all = counters[6];
    // wave-aggregated cursors: one atomic per wave and list instead of one per key (same-address atomics serialise)
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned int stride = gridDim.x * blockDim.x;
    for (unsigned int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += stride) {      // wave-uniform trip count
        const unsigned int i = i0 + threadIdx.x;
        const bool live = i < n;
        const unsigned long long key = live ? cand[i] : 0ull;
        const unsigned int bin = key_bin(key);
        const bool to_sel = live && (all || bin > bstar), to_bnd = live && !to_sel && bin == bstar;
        const unsigned long long ms = __ballot(to_sel), mb = __ballot(to_bnd);
        unsigned int base_s = 0, base_b = 0;
        if (lane == 0) {
            if (ms) base_s = atomicAdd(&counters[2], (unsigned int)__popcll(ms));
            if (mb) base_b = atomicAdd(&counters[3], (unsigned int)__popcll(mb));
        }
        base_s = __shfl(base_s, 0);
        base_b = __shfl(base_b, 0);
        if (to_sel) {
            const unsigned int pos = base_s + (unsigned int)__popcll(ms & below);
            if (pos < (unsigned int)sel_cap) sel[pos] = key;
        } else if (to_bnd) {
            __hip_atomic_store(bnd + base_b + (unsigned int)__popcll(mb & below), key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bnd has cand_cap entries
        }
    }
    // the last block to finish ranks the boundary bin (counters[8] = ticket): one launch less.  The boundary keys are
    // written and read with device-scope atomic stores / loads and the cursors are atomics: no fence (see sfd2_ld_agent).
    __shared__ unsigned int s_last;
    SFD2_BARRIER_DRAIN();   // every store / atomic of this block is acknowledged: the vmcnt(0) is written out -- a
                            // workgroup-scope __syncthreads() does not wait for global operations on this target
    if (threadIdx.x == 0) s_last = atomicAdd(&counters[8], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    const unsigned int nb = sfd2_ld_agent(counters + 3), need = counters[5], k = counters[1];
    if (counters[6] || need == 0) return;
    __shared__ unsigned long long tile[NT];
    const unsigned int base_out = k - need;
    for (unsigned int i0 = 0; i0 < nb; i0 += blockDim.x) {
        const unsigned int i = i0 + threadIdx.x;
        const unsigned long long mine = i < nb ? __hip_atomic_load(bnd + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        unsigned int rank = 0;
        for (unsigned int b = 0; b < nb; b += NT) {
            tile[threadIdx.x] = (b + threadIdx.x < nb) ? __hip_atomic_load(bnd + b + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            __syncthreads();
            const unsigned int lim = nb - b < NT ? nb - b : NT;
            for (unsigned int t = 0; t < lim; ++t) rank += tile[t] > mine ? 1u : 0u;
            __syncthreads();
        }
        if (i < nb && rank < need && base_out + rank < (unsigned int)sel_cap) sel[base_out + rank] = mine;
    }
}

// rank-by-counting sort of the selected keys (unique) into descending order.  (A single-block bitonic network over 4 096
// keys in LDS -- 78 compare-exchange passes -- was measured: the chain 24 -> 50 us.  Not kept.)
// block = 16 keys x 16 slices of the comparison range (256 blocks for 4096 keys instead of 64: the kernel is
// latency-bound, 20 -> see profiles/ us); every slice streams its part of the keys through LDS in 256-key tiles.
#define RS_KEYS 16
#define RS_SL 16
__global__ __launch_bounds__(NT)
void rank_sort_kernel(const unsigned long long *__restrict__ sel, unsigned long long *__restrict__ sorted,
                      int sel_cap, const unsigned int *__restrict__ counters, int W, float *__restrict__ kpts,
                      float *__restrict__ scores)
{
    __shared__ unsigned long long tile[RS_SL][64];
    __shared__ unsigned int part[RS_SL][RS_KEYS];
    unsigned int n = counters[1];
    if (n > (unsigned int)sel_cap) n = sel_cap;
    if (blockIdx.x * (unsigned int)RS_KEYS >= n) return;
    const int li = threadIdx.x & (RS_KEYS - 1), q = threadIdx.x / RS_KEYS;      // key within the block, slice
    const unsigned int i = blockIdx.x * RS_KEYS + li;
    const unsigned long long mine = i < n ? sel[i] : 0ull;
    const unsigned int qlen = (n + RS_SL - 1) / RS_SL, j0 = q * qlen;
    const unsigned int j1 = j0 + qlen < n ? j0 + qlen : n;
    unsigned int rank = 0;
    // this thread's four keys of the first four tiles (4 096 selected keys = four tiles per slice) are requested up front:
    // one memory round trip instead of one per tile
    unsigned long long pre[4][64 / RS_KEYS];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int tt = 0; tt < 64 / RS_KEYS; ++tt) {
            const unsigned int j = j0 + pb * 64 + li + tt * RS_KEYS;
            pre[pb][tt] = (pb * 64u < qlen && j < j1) ? sel[j] : 0ull;
        }
    for (unsigned int b = 0; b < qlen; b += 64) {            // same trip count for every slice
        if (b < 256) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
                if (b == pb * 64u)
#pragma unroll
                    for (int tt = 0; tt < 64 / RS_KEYS; ++tt) tile[q][li + tt * RS_KEYS] = pre[pb][tt];
        } else {
            for (int t = li; t < 64; t += RS_KEYS) tile[q][t] = (j0 + b + t < j1) ? sel[j0 + b + t] : 0ull;
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < 64; ++t) rank += tile[q][t] > mine ? 1u : 0u;
        __syncthreads();
    }
    part[q][li] = rank;
    __syncthreads();
    if (q == 0 && i < n) {
        unsigned int r = 0;
#pragma unroll
        for (int k = 0; k < RS_SL; ++k) r += part[k][li];
        sorted[r] = mine;
        if (kpts) {   // selection path: the key's pixel index and score go straight to the output row
            const unsigned int idx = 0xFFFFFFFFu - (unsigned int)(mine & 0xFFFFFFFFull);
            kpts[2 * r] = (float)(idx % (unsigned int)W);
            kpts[2 * r + 1] = (float)(idx / (unsigned int)W);
            scores[r] = __uint_as_float((unsigned int)(mine >> 32));
        }
    }
}

void launch_topk_sort(hipStream_t st, bool threshold_done, const unsigned long long *cand, int cand_cap, int top_k,
                      unsigned long long *sel, unsigned long long *sorted, int sel_cap, unsigned int *counters,
                      unsigned long long *bnd, int W, float *kpts, float *scores)
{
    if (!threshold_done) hipLaunchKernelGGL(select_threshold_kernel, dim3(1), dim3(1024), 0, st, cand_cap, top_k, counters);
    int grid = (cand_cap + NT - 1) / NT;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(compact_selected_kernel, dim3(grid), dim3(NT), 0, st, cand, cand_cap, sel, sel_cap, bnd, counters);
    hipLaunchKernelGGL(rank_sort_kernel, dim3((sel_cap + RS_KEYS - 1) / RS_KEYS), dim3(NT), 0, st, sel, sorted, sel_cap, counters, W,
                       kpts, scores);
}

__global__ __launch_bounds__(NT)
void keys_to_kpts_kernel(const unsigned long long *__restrict__ sorted, const unsigned int *__restrict__ counters,
                         int W, float *__restrict__ kpts, float *__restrict__ scores, int cap)
{
    unsigned int n = counters[1];
    if (n > (unsigned int)cap) n = cap;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = sorted[i];
    const unsigned int idx = 0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull);
    kpts[2 * i] = (float)(idx % (unsigned int)W);
    kpts[2 * i + 1] = (float)(idx / (unsigned int)W);
    scores[i] = __uint_as_float((unsigned int)(key >> 32));
}

void launch_keys_to_kpts(hipStream_t st, const unsigned long long *sorted, const unsigned int *counters, int W,
                         float *kpts, float *scores, int cap)
{
    if (cap <= 0) return;
    hipLaunchKernelGGL(keys_to_kpts_kernel, dim3((cap + NT - 1) / NT), dim3(NT), 0, st, sorted, counters, W, kpts, scores, cap);
}

// ---------------------------------------------------------------- greedy grid NMS (extract.py:17-84 nms_fast)
// The reference visits candidates in score order and keeps one iff no kept candidate lies within
// Chebyshev distance `dist`.  Exact parallel form: a candidate is KEPT once every higher-priority
// candidate in its window is SUPPRESSED, SUPPRESSED once any candidate in its window is KEPT;
// iterate (double-buffered states) to the fixed point -- identical to the sequential result.
// Priority = 64-bit key (score bits, then lower pixel index first), 0 = not a candidate.
#define GS_NONE 0
#define GS_UNDECIDED 1
#define GS_KEPT 2
#define GS_SUPPRESSED 3
__global__ __launch_bounds__(NT)
void greedy_init_kernel(const float *__restrict__ heat, int n, float conf_th, unsigned long long *__restrict__ keys,
                        unsigned char *__restrict__ state)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = heat[i];
    const bool c = v >= conf_th;                      // np.where(heatmap >= conf_thresh)  (extract.py:230)
    keys[i] = c ? (((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i)) : 0ull;
    state[i] = c ? GS_UNDECIDED : GS_NONE;
}

__global__ __launch_bounds__(NT)
void greedy_iter_kernel(const unsigned long long *__restrict__ keys, const unsigned char *__restrict__ sin,
                        unsigned char *__restrict__ sout, int H, int W, int dist, unsigned int *__restrict__ undecided)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int i = y * W + x;
    const unsigned char st = sin[i];
    if (st != GS_UNDECIDED) { sout[i] = st; return; }
    const unsigned long long mine = keys[i];
    bool any_kept = false, blocked = false;
    const int y0 = max(0, y - dist), y1 = min(H - 1, y + dist), x0 = max(0, x - dist), x1 = min(W - 1, x + dist);
    for (int yy = y0; yy <= y1; ++yy)
        for (int xx = x0; xx <= x1; ++xx) {
            const int j = yy * W + xx;
            const unsigned char sj = sin[j];
            if (sj == GS_KEPT) any_kept = true;
            else if (sj == GS_UNDECIDED && keys[j] > mine) blocked = true;
        }
    unsigned char ns = GS_UNDECIDED;
    if (any_kept) ns = GS_SUPPRESSED;
    else if (!blocked) ns = GS_KEPT;
    sout[i] = ns;
    if (ns == GS_UNDECIDED) atomicAdd(undecided, 1u);
}

__global__ __launch_bounds__(NT)
void greedy_final_kernel(const float *__restrict__ heat, const unsigned char *__restrict__ state, int n, float *__restrict__ kept)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) kept[i] = state[i] == GS_KEPT ? heat[i] : 0.0f;
}

void launch_greedy_init(hipStream_t st, const float *heat, int n, float conf_th, unsigned long long *keys, unsigned char *state)
{
    hipLaunchKernelGGL(greedy_init_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, st, heat, n, conf_th, keys, state);
}
void launch_greedy_iter(hipStream_t st, const unsigned long long *keys, const unsigned char *sin, unsigned char *sout,
                        int H, int W, int dist, unsigned int *undecided)
{
    hipLaunchKernelGGL(greedy_iter_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(NT), 0, st, keys, sin, sout, H, W, dist, undecided);
}
void launch_greedy_final(hipStream_t st, const float *heat, const unsigned char *state, int n, float *kept)
{
    hipLaunchKernelGGL(greedy_final_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, st, heat, state, n, kept);
}

// ---------------------------------------------------------------- descriptor sampling
// One wave per key point, lane l owns channels 2l, 2l+1.  The four taps are L2-normalised on
// the fly (F.normalize of the dense map, nets/sfd2.py:342, commutes with the gather), blended
// with torch's grid_sample weights (zeros padding, align_corners=False) and re-normalised.
// torch grid_sample geometry of one key point (zeros padding, align_corners=False): corner pixels, validity, weights
struct SampleGeom {
    int x0, y0, x1, y1;
    bool vx0, vx1, vy0, vy1;
    float w_nw, w_ne, w_sw, w_se;
};
__device__ __forceinline__ SampleGeom sample_geom(float kx, float ky, float half_w, float half_h, int hc, int wc)
{
    SampleGeom g;
    const float gx = __fsub_rn(__fdiv_rn(kx, half_w), 1.0f);
    const float gy = __fsub_rn(__fdiv_rn(ky, half_h), 1.0f);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)wc), 1.0f), 2.0f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)hc), 1.0f), 2.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    g.x0 = (int)fx; g.y0 = (int)fy; g.x1 = g.x0 + 1; g.y1 = g.y0 + 1;
    const float ex = __fsub_rn(__fadd_rn(fx, 1.0f), ix), ey = __fsub_rn(__fadd_rn(fy, 1.0f), iy);
    const float dx = __fsub_rn(ix, fx), dy = __fsub_rn(iy, fy);
    g.w_nw = __fmul_rn(ex, ey); g.w_ne = __fmul_rn(dx, ey); g.w_sw = __fmul_rn(ex, dy); g.w_se = __fmul_rn(dx, dy);
    g.vx0 = g.x0 >= 0 && g.x0 < wc; g.vx1 = g.x1 >= 0 && g.x1 < wc;
    g.vy0 = g.y0 >= 0 && g.y0 < hc; g.vy1 = g.y1 >= 0 && g.y1 < hc;
    return g;
}

__global__ __launch_bounds__(NT)
void sample_desc_kernel(const float *__restrict__ dmap, int hc, int wc, float half_w, float half_h,
                        const float *__restrict__ kpts, const unsigned int *__restrict__ count, int n_max,
                        float *__restrict__ out, int compact /* dmap = [key point][corner][128], not the dense map */)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    int n = n_max;
    if (count) { const unsigned int c = *count; if ((unsigned int)n > c) n = (int)c; }
    if (i >= n) return;
    const SampleGeom g = sample_geom(kpts[2 * i], kpts[2 * i + 1], half_w, half_h, hc, wc);
    float a0 = 0.0f, a1 = 0.0f;
#define SFD2_TAP(valid_, yy_, xx_, wgt_, corner_)                                                    \
    if (valid_) {                                                                                    \
        const size_t row = compact ? (size_t)i * 4 + (corner_) : (size_t)(yy_) * wc + (xx_);         \
        const float2 v = *reinterpret_cast<const float2 *>(dmap + row * 128 + 2 * lane);             \
        const float nrm = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y)), 1e-12f);                     \
        a0 += __fdiv_rn(v.x, nrm) * (wgt_);                                                          \
        a1 += __fdiv_rn(v.y, nrm) * (wgt_);                                                          \
    }
    SFD2_TAP(g.vy0 && g.vx0, g.y0, g.x0, g.w_nw, 0)
    SFD2_TAP(g.vy0 && g.vx1, g.y0, g.x1, g.w_ne, 1)
    SFD2_TAP(g.vy1 && g.vx0, g.y1, g.x0, g.w_sw, 2)
    SFD2_TAP(g.vy1 && g.vx1, g.y1, g.x1, g.w_se, 3)
#undef SFD2_TAP
    const float nrm = sqrtf(wave_sum(a0 * a0 + a1 * a1));
    *reinterpret_cast<float2 *>(out + (size_t)i * 128 + 2 * lane) = make_float2(__fdiv_rn(a0, nrm), __fdiv_rn(a1, nrm));
}

void launch_sample_desc(hipStream_t st, const float *dmap, int hc, int wc, int nh, int nw, const float *kpts,
                        const unsigned int *count, int n_max, float *out, int compact)
{
    if (n_max <= 0) return;
    hipLaunchKernelGGL(sample_desc_kernel, dim3((n_max + 3) / 4), dim3(NT), 0, st, dmap, hc, wc, (float)nw / 2.0f,
                       (float)nh / 2.0f, kpts, count, n_max, out, compact);
}

// Sparse descriptor head (extract path): convDb is a 1x1 convolution and only the four bilinear corners of the selected
// key points are ever sampled (16 384 of 120 000 pixels at 1600x1200, top-4096), so on the throughput path convDb runs
// AFTER the selection, on those pixels only, and the 61 MB fp32 descriptor map is never written.  Same-box A/B
// (tools/ab_option.py sparse_desc 0 1): 1.135 -> 1.083 ms per extract -- 52 us, of which only 18 are the kernels' own
// (dense convDb 28 + sampling 9.5 vs 19): the rest is what the map's write cost the kernels after it.
// The three steps in ONE kernel: a block takes 8 key points, gathers their 64 corner pixels (256 fp16 channels each) into
// LDS, runs convDb on them with MFMAs (128 out channels x 64 pixels x K = 256; wave w owns channels 32w .. 32w + 31, its
// filter fragments come straight from the packed filters in L2), parks the fp32 result in LDS and samples it.  K ascends
// in 16-wide slices into one accumulator and the epilogue is acc * scale + shift, exactly as conv_igemm2 computes the
// dense map, so the descriptors are bit-identical to the dense path's.
#define DH_KP 8         // key points per block: 32 corner records = one MFMA tile; 512 blocks for 4 096 key points, two per CU
                        // (16 per block: 17.5 us, 8: 12.7 us -- the kernel is a chain of memory round trips, concurrency hides them)
#define DH_XREC 528      // bytes per gathered pixel record: 512 + 16 pad (conflict-free ds_read_b128)
#define DH_OREC 132      // floats per conv output record: 128 + 4 pad
__global__ __launch_bounds__(NT)
void desc_head_kernel(const half_t *__restrict__ fmap /*[hc][wc][256]*/, int hc, int wc, float half_w, float half_h,
                      const half_t *__restrict__ wpk /*[8 chunks][CoutP][32]*/, int CoutP, const float *__restrict__ scale,
                      const float *__restrict__ shift, const float *__restrict__ kpts, const unsigned int *__restrict__ count,
                      int n_max, float *__restrict__ out, int compact /* fmap = [key point][corner][256] (sparse_da3_kernel), not the dense map */)
{
    __shared__ __attribute__((aligned(16))) unsigned char X[4 * DH_KP * DH_XREC];
    __shared__ __attribute__((aligned(16))) float O[4 * DH_KP * DH_OREC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n = n_max;
    if (count) { const unsigned int c = *count; if ((unsigned int)n > c) n = (int)c; }
    const int k0 = blockIdx.x * DH_KP;
    if (k0 >= n) return;

    // filter fragments of this wave's 32 output channels (requested first: they land while the corners are gathered)
    h8_t a[16];
    const int lrow = lane & 31, lhi = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        a[kk] = *reinterpret_cast<const h8_t *>(wpk + ((size_t)(kk >> 1) * CoutP + wave * 32 + lrow) * 32 + (kk & 1) * 16 + lhi * 8);

    // gather: half a wave per pixel record (32 lanes x 16 B), zeros for corners outside the map / key points past the count.
    // Written as three unrolled stages (coordinates, map loads, LDS stores) with unconditional loads from a clamped
    // address: as one loop with the loads under their conditions, the eight records of a half-wave were eight
    // back-to-back memory round trips (coordinate, then pixel), most of this kernel's 19 us.
    constexpr int DH_NR = 4 * DH_KP / 8;
    float kx[DH_NR], ky[DH_NR];
#pragma unroll
    for (int i = 0; i < DH_NR; ++i) {
        const int kp = k0 + ((wave * 2 + lhi + 8 * i) >> 2);
        const int kc = kp < n ? kp : n - 1;
        kx[i] = kpts[2 * kc];
        ky[i] = kpts[2 * kc + 1];
    }
    uint4 gv[DH_NR];
    bool gok[DH_NR];
#pragma unroll
    for (int i = 0; i < DH_NR; ++i) {
        const int rec = wave * 2 + lhi + 8 * i;
        const int kp = k0 + (rec >> 2), corner = rec & 3;
        const SampleGeom g = sample_geom(kx[i], ky[i], half_w, half_h, hc, wc);
        const int yy = (corner & 2) ? g.y1 : g.y0, xx = (corner & 1) ? g.x1 : g.x0;
        gok[i] = kp < n && ((corner & 2) ? g.vy1 : g.vy0) && ((corner & 1) ? g.vx1 : g.vx0);
        const size_t pix = gok[i] ? (compact ? (size_t)kp * 4 + corner : (size_t)yy * wc + xx) : 0;
        gv[i] = *reinterpret_cast<const uint4 *>(fmap + pix * 256 + lrow * 8);
    }
#pragma unroll
    for (int i = 0; i < DH_NR; ++i) {
        const int rec = wave * 2 + lhi + 8 * i;
        *reinterpret_cast<uint4 *>(X + rec * DH_XREC + lrow * 16) = gok[i] ? gv[i] : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    constexpr int DH_NT = 4 * DH_KP / 32;             // 32-record MFMA tiles
    f32x16_t acc[DH_NT];
#pragma unroll
    for (int t = 0; t < DH_NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int t = 0; t < DH_NT; ++t) {
            const h8_t b = *reinterpret_cast<const h8_t *>(X + (t * 32 + lrow) * DH_XREC + kk * 32 + lhi * 16);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], b, acc[t], 0, 0, 0);
        }
    // C layout: lane owns pixel (t * 32 + lrow), channels wave * 32 + 8 q + 4 lhi + j
#pragma unroll
    for (int t = 0; t < DH_NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c0 = wave * 32 + 8 * q + 4 * lhi;
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c0);
            const float4 sh = *reinterpret_cast<const float4 *>(shift + c0);
            const float4 v = make_float4(acc[t][4 * q + 0] * sc.x + sh.x, acc[t][4 * q + 1] * sc.y + sh.y,
                                         acc[t][4 * q + 2] * sc.z + sh.z, acc[t][4 * q + 3] * sc.w + sh.w);
            *reinterpret_cast<float4 *>(O + (t * 32 + lrow) * DH_OREC + c0) = v;
        }
    __syncthreads();

    // sampling: one wave per key point, lane l owns channels 2l, 2l + 1 (as sample_desc_kernel)
    for (int j = wave; j < DH_KP; j += NT / 64) {
        const int kp = k0 + j;
        if (kp >= n) break;
        const SampleGeom g = sample_geom(kpts[2 * kp], kpts[2 * kp + 1], half_w, half_h, hc, wc);
        float a0 = 0.0f, a1 = 0.0f;
#define SFD2_TAP(valid_, wgt_, corner_)                                                              \
        if (valid_) {                                                                                \
            const float2 v = *reinterpret_cast<const float2 *>(O + (4 * j + (corner_)) * DH_OREC + 2 * lane); \
            const float nrm = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y)), 1e-12f);                 \
            a0 += __fdiv_rn(v.x, nrm) * (wgt_);                                                      \
            a1 += __fdiv_rn(v.y, nrm) * (wgt_);                                                      \
        }
        SFD2_TAP(g.vy0 && g.vx0, g.w_nw, 0)
        SFD2_TAP(g.vy0 && g.vx1, g.w_ne, 1)
        SFD2_TAP(g.vy1 && g.vx0, g.w_sw, 2)
        SFD2_TAP(g.vy1 && g.vx1, g.w_se, 3)
#undef SFD2_TAP
        const float nrm = sqrtf(wave_sum(a0 * a0 + a1 * a1));
        *reinterpret_cast<float2 *>(out + (size_t)kp * 128 + 2 * lane) = make_float2(__fdiv_rn(a0, nrm), __fdiv_rn(a1, nrm));
    }
}

void launch_desc_head(hipStream_t st, const half_t *fmap, int hc, int wc, int nh, int nw, const half_t *wpk, int CoutP,
                      const float *scale, const float *shift, const float *kpts, const unsigned int *count, int n_max, float *out, int compact)
{
    if (n_max <= 0) return;
    hipLaunchKernelGGL(desc_head_kernel, dim3((n_max + DH_KP - 1) / DH_KP), dim3(NT), 0, st, fmap, hc, wc, (float)nw / 2.0f,
                       (float)nh / 2.0f, wpk, CoutP, scale, shift, kpts, count, n_max, out, compact);
}

// ---------------------------------------------------------------- descriptors as the reference STORES them (SFD2_FLAG_DESC_STORE64)
// extract_localization.py:253 transposes the [N,128] descriptors to [128,N] and :269-272 writes them as float64: on a host that feeds several GPUs the
// cast + transposing copy of 4096 x 128 values per image is the writer threads' largest cost (tools/host_soak.py).  Here the device does both while the
// descriptors are still in HBM: in [n_max][128] fp32 -> out [128][pitch] fp64 (exact: every fp32 is an fp64), columns >= count zero.
__global__ __launch_bounds__(NT)
void desc_store64_kernel(const float *__restrict__ in, const unsigned int *__restrict__ count, int n_max, double *__restrict__ out, int pitch)
{
    __shared__ float tile[64][129];
    int n = n_max;
    if (count) { const unsigned int c = *count; if ((unsigned int)n > c) n = (int)c; }
    const int k0 = blockIdx.x * 64, tid = threadIdx.x;
    for (int i = tid; i < 64 * 128; i += NT) {
        const int r = i >> 7, ch = i & 127;
        tile[r][ch] = (k0 + r < n) ? in[(size_t)(k0 + r) * 128 + ch] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 128; i += NT) {
        const int ch = i >> 6, r = i & 63;
        if (k0 + r < pitch) out[(size_t)ch * pitch + k0 + r] = (double)tile[r][ch];
    }
}

void launch_desc_store64(hipStream_t st, const float *in, const unsigned int *count, int n_max, double *out, int pitch)
{
    if (pitch <= 0) return;
    hipLaunchKernelGGL(desc_store64_kernel, dim3((pitch + 63) / 64), dim3(NT), 0, st, in, count, n_max, out, pitch);
}

// ---------------------------------------------------------------- dense descriptor normalise + NHWC -> NCHW
__global__ __launch_bounds__(NT)
void desc_normalise_nchw_kernel(const float *__restrict__ in, int npix, float *__restrict__ out)
{
    __shared__ float tile[64][129];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p0 = blockIdx.x * 64;
    for (int r = wave; r < 64; r += 4) {
        const int p = p0 + r;
        float2 v = make_float2(0.0f, 0.0f);
        if (p < npix) v = *reinterpret_cast<const float2 *>(in + (size_t)p * 128 + 2 * lane);
        const float nrm = fmaxf(sqrtf(wave_sum(v.x * v.x + v.y * v.y)), 1e-12f);
        tile[r][2 * lane] = __fdiv_rn(v.x, nrm);
        tile[r][2 * lane + 1] = __fdiv_rn(v.y, nrm);
    }
    __syncthreads();
    for (int c = wave; c < 128; c += 4) {
        const int p = p0 + lane;
        if (p < npix) out[(size_t)c * npix + p] = tile[lane][c];
    }
}

void launch_desc_normalise_nchw(hipStream_t st, const float *in, int npix, float *out)
{
    hipLaunchKernelGGL(desc_normalise_nchw_kernel, dim3((npix + 63) / 64), dim3(NT), 0, st, in, npix, out);
}

// ---------------------------------------------------------------- scale pyramid (nets/extractor.py:118-124,211-236)
// level image = F.interpolate(norm_RGB(img), (nh, nw), bilinear, align_corners=False): normalise the four taps with
// norm_RGB's IEEE sub + div, blend with the pinned torch rounding sequence (lin_coef / bilerp above).
__device__ __forceinline__ float norm_tap(const float *__restrict__ img, int mode, int c, size_t plane, int W, int y, int x)
{
    float v;
    if (mode & 2) {
        const int cs = (mode & 4) ? 2 - c : c;
        v = __fdiv_rn((float)reinterpret_cast<const unsigned char *>(img)[((size_t)y * W + x) * 3 + cs], 255.0f);
    } else {
        v = img[c * plane + (size_t)y * W + x];
    }
    if (mode & 1) {
        const float m = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
        const float sd = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
        v = __fdiv_rn(__fsub_rn(v, m), sd);
    }
    return v;
}

__global__ __launch_bounds__(NT)
void norm_resize_kernel(const float *__restrict__ img, int mode, int H, int W, int nh, int nw, float sc_y, float sc_x,
                        float *__restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int c = blockIdx.z;
    if (x >= nw || y >= nh) return;
    const LinCoef cy = lin_coef(y, H, nh, sc_y), cx = lin_coef(x, W, nw, sc_x);
    const size_t plane = (size_t)H * W;
    const float v00 = norm_tap(img, mode, c, plane, W, cy.i0, cx.i0), v01 = norm_tap(img, mode, c, plane, W, cy.i0, cx.i1);
    const float v10 = norm_tap(img, mode, c, plane, W, cy.i1, cx.i0), v11 = norm_tap(img, mode, c, plane, W, cy.i1, cx.i1);
    const float top = __fmaf_rn(v00, cx.l0, __fmul_rn(v01, cx.l1));
    const float bot = __fmaf_rn(v10, cx.l0, __fmul_rn(v11, cx.l1));
    out[(size_t)c * nh * nw + (size_t)y * nw + x] = __fmaf_rn(top, cy.l0, __fmul_rn(bot, cy.l1));
}

void launch_norm_resize(hipStream_t st, const float *img, int mode, int H, int W, int nh, int nw, float *out)
{
    hipLaunchKernelGGL(norm_resize_kernel, dim3((nw + 63) / 64, (nh + 3) / 4, 3), dim3(NT), 0, st, img, mode, H, W, nh, nw,
                       (float)H / (float)nh, (float)W / (float)nw, out);
}

// one pyramid level's selected key points -> the staging arrays: x * W / nw, y * H / nh in fp32 (two roundings each,
// nets/extractor.py:211-212), scores copied, the level's count recorded.
__global__ __launch_bounds__(NT)
void ms_append_kernel(const float *__restrict__ kpts, const float *__restrict__ scores, const unsigned int *__restrict__ count,
                      int cap, float W, float nw, float H, float nh, float *__restrict__ kp_out, float *__restrict__ sc_out,
                      unsigned int *__restrict__ level_count)
{
    unsigned int n = *count;
    if (n > (unsigned int)cap) n = cap;
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *level_count = n;
    if (i >= n) return;
    kp_out[2 * i] = __fdiv_rn(__fmul_rn(kpts[2 * i], W), nw);
    kp_out[2 * i + 1] = __fdiv_rn(__fmul_rn(kpts[2 * i + 1], H), nh);
    sc_out[i] = scores[i];
}

void launch_ms_append(hipStream_t st, const float *kpts, const float *scores, const unsigned int *count, int cap, int W, int nw,
                      int H, int nh, float *kp_out, float *sc_out, unsigned int *level_count)
{
    hipLaunchKernelGGL(ms_append_kernel, dim3((cap + NT - 1) / NT > 0 ? (cap + NT - 1) / NT : 1), dim3(NT), 0, st, kpts, scores,
                       count, cap, (float)W, (float)nw, (float)H, (float)nh, kp_out, sc_out, level_count);
}

// concatenation order position -> 64-bit key (score bits, then earlier position first); ms_counters[1] = total
struct MsLevels { int n_levels; int offset[8]; };   // staging offset of each level (host-known capacities)

__device__ __forceinline__ void ms_locate(const MsLevels &lv, const unsigned int *__restrict__ level_count, unsigned int pos,
                                          int *level, unsigned int *idx)
{
    unsigned int base = 0;
    int l = 0;
    for (; l < lv.n_levels - 1; ++l) {
        const unsigned int n = level_count[l];
        if (pos < base + n) break;
        base += n;
    }
    *level = l;
    *idx = pos - base;
}

__global__ __launch_bounds__(NT)
void ms_keys_kernel(MsLevels lv, const unsigned int *__restrict__ level_count, const float *__restrict__ sc_stage,
                    unsigned long long *__restrict__ keys, int cap_total, unsigned int *__restrict__ ms_counters)
{
    unsigned int total = 0;
    for (int l = 0; l < lv.n_levels; ++l) total += level_count[l];
    const unsigned int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos == 0) ms_counters[1] = total;
    if (pos >= total || pos >= (unsigned int)cap_total) return;
    int l; unsigned int i;
    ms_locate(lv, level_count, pos, &l, &i);
    const float s = sc_stage[lv.offset[l] + i];
    keys[pos] = ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - pos);
}

// output row r <- staged key point of sorted[r] (or of position r when sorted == nullptr: top_k <= 0 keeps the
// concatenation order, nets/extractor.py:322-330).  One wave per key point (128-float descriptor row).
__global__ __launch_bounds__(NT)
void ms_gather_kernel(MsLevels lv, const unsigned int *__restrict__ level_count, const unsigned long long *__restrict__ sorted,
                      const float *__restrict__ kp_stage, const float *__restrict__ sc_stage, const float *__restrict__ de_stage,
                      int n_max, float *__restrict__ kp_out, float *__restrict__ sc_out, float *__restrict__ de_out,
                      unsigned int *__restrict__ ms_counters)
{
    unsigned int total = 0;
    for (int l = 0; l < lv.n_levels; ++l) total += level_count[l];
    unsigned int n = total < (unsigned int)n_max ? total : (unsigned int)n_max;
    const int lane = threadIdx.x & 63;
    const unsigned int r = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (r == 0 && lane == 0) ms_counters[2] = n;
    if (r >= n) return;
    const unsigned int pos = sorted ? 0xFFFFFFFFu - (unsigned int)(sorted[r] & 0xFFFFFFFFull) : r;
    int l; unsigned int i;
    ms_locate(lv, level_count, pos, &l, &i);
    const size_t src = (size_t)lv.offset[l] + i;
    if (lane < 2) kp_out[2 * (size_t)r + lane] = kp_stage[2 * src + lane];
    if (lane == 2) sc_out[r] = sc_stage[src];
    if (de_out)
        *reinterpret_cast<float2 *>(de_out + (size_t)r * 128 + 2 * lane) =
            *reinterpret_cast<const float2 *>(de_stage + src * 128 + 2 * lane);
}

void launch_ms_merge(hipStream_t st, int n_levels, const int *offsets, const unsigned int *level_count, const float *kp_stage,
                     const float *sc_stage, const float *de_stage, int cap_total, int top_k, unsigned long long *keys,
                     unsigned long long *sorted, unsigned int *ms_counters, int n_max, float *kp_out, float *sc_out, float *de_out)
{
    MsLevels lv;
    lv.n_levels = n_levels;
    for (int l = 0; l < 8; ++l) lv.offset[l] = l < n_levels ? offsets[l] : 0;
    const unsigned long long *order = nullptr;
    if (top_k > 0) {
        hipLaunchKernelGGL(ms_keys_kernel, dim3((cap_total + NT - 1) / NT), dim3(NT), 0, st, lv, level_count, sc_stage, keys,
                           cap_total, ms_counters);
        hipLaunchKernelGGL(rank_sort_kernel, dim3((cap_total + RS_KEYS - 1) / RS_KEYS), dim3(NT), 0, st, keys, sorted, cap_total, ms_counters, 1,
                           nullptr, nullptr);
        order = sorted;
    }
    if (n_max <= 0) return;
    hipLaunchKernelGGL(ms_gather_kernel, dim3((n_max + 3) / 4), dim3(NT), 0, st, lv, level_count, order, kp_stage, sc_stage,
                       de_stage, n_max, kp_out, sc_out, de_out, ms_counters);
}
