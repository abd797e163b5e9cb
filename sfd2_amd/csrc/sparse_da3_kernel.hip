// convDa.3 (3x3, 256 -> 256, no activation: nets/sfd2.py:340-342) on the SAMPLED pixels only.  extract_resnet_return samples the
// unit-norm descriptor map at the selected key points (nets/extractor.py:199-208): bilinear, i.e. four corner pixels of the
// 1/4-resolution map per key point.  convDb (1x1) already runs on those corners only (desc_head_kernel, post_kernels.hip); its
// input, convDa.3's output, is needed at the same 4 x K pixels -- 16 384 of 120 000 at 1600x1200 / top-4096 -- so on the
// extract path this kernel computes convDa.3 there and nowhere else: a key point's four corners are a 2 x 2 block whose 3x3
// neighbourhoods are one 4 x 4 patch of convDa.0's output.
//
//   block   16 key points (64 output pixels = two 32-pixel MFMA tiles) x 128 output channels (grid.y = 2); 4 waves, a wave owns
//           32 channels; up to four blocks per CU, which is what hides the gather round trips
//   K loop  four 64-channel chunks: the 16 patches' records (16 x 16 pixels x 128 B = 32 KB) are copied global -> LDS
//           (a record's eight 16-byte parts at slot part ^ (record & 7) ^ (key point & 3): ds_read_b128 serves a wave in four groups of sixteen
//           lanes = four key points x four corners, whose sixteen 16-byte slots must differ across the 256-byte bank row -- the corners differ through
//           the record term, the group's key points {0, 3, 5, 6} / {1, 2, 4, 7} through theirs; without the second term every read was four-way
//           conflicted: SQ_LDS_BANK_CONFLICT 73 % of SQ_LDS_IDX_ACTIVE, profiles/r05s_pmc_summary.txt), then
//           9 taps x 4 K-slices of v_mfma_f32_32x32x16_f16 per tile, filter fragments straight from the packed filters in L2
//   output  compact [key point][corner][256] fp16 = what desc_head_kernel gathers from the dense map otherwise
//
// Same products as the dense layer (conv3x3_pp), fp32 summation order differs (chunk-major there too, taps inside): outputs
// agree to fp32 rounding before the fp16 store, i.e. are identical except where a sum sits on an fp16 rounding boundary.
#include "sfd2_internal.h"

#define SD_KP 16
#define SD_NT 256
// -DSFD2_SD_PF=0: the filter fragments as first written (requested per tap, three blocks per CU) -- the A side of profiles/r05v_sparse_da3_ab.txt
#ifndef SFD2_SD_PF
#define SFD2_SD_PF 1
#endif
// fragment sets in the prefetch ring (SD_RING - 1 taps in flight behind the one in use); must divide the 36 taps of the four chunks, whose
// loop is unrolled so that the ring index stays a constant (6: seven spilled registers, slower)
#ifndef SFD2_SD_RING
#define SFD2_SD_RING 4
#endif
#ifndef SFD2_SD_RING_X3      // the three-pass form holds twice the fragments per tap and twice the accumulators (1 = no prefetch)
#define SFD2_SD_RING_X3 2
#endif
#define SD_XB (SD_KP * 16 * 128)

typedef __attribute__((address_space(3))) void sd_lds_t;
typedef const __attribute__((address_space(1))) void sd_gbl_t;

// nets/extractor.py:199-208 (F.grid_sample, bilinear, align_corners=False): the top-left corner of the key point's 2 x 2 block,
// the arithmetic of post_kernels.hip's sample_geom
__device__ __forceinline__ void sd_corner(float kx, float ky, float half_w, float half_h, int hc, int wc, int &x0, int &y0)
{
    const float gx = __fsub_rn(__fdiv_rn(kx, half_w), 1.0f);
    const float gy = __fsub_rn(__fdiv_rn(ky, half_h), 1.0f);
    const float ix = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gx, 1.0f), (float)wc), 1.0f), 2.0f);
    const float iy = __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(gy, 1.0f), (float)hc), 1.0f), 2.0f);
    x0 = (int)floorf(ix);
    y0 = (int)floorf(iy);
}

// X3 (SFD2_PREC_F16X3): fmap / fmap_lo = convDa.0's output as hi / lo' planes, wpk = [hi chunks][lo' chunks], three MFMAs per K
// slice into two accumulators (hi x hi; hi x lo' + lo' x hi, weighted 2^-11), fp32 output [n_max][4][256]
// LINA: wpk is the repacked array (sparse_da3_repack_kernel below)
template <bool X3, bool LINA = false>
__global__ __launch_bounds__(SD_NT, (X3 || SFD2_SD_PF) ? 2 : 3)
void sparse_da3_kernel(const half_t *__restrict__ fmap /*convDa.0's output [hc][wc][256]*/, const half_t *__restrict__ fmap_lo, int hc, int wc,
                       float half_w, float half_h,
                       const half_t *__restrict__ wpk /*[8 chunks of 32][9 taps][CoutP][32]*/, int CoutP,
                       const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                       const float *__restrict__ kpts, const unsigned int *__restrict__ count, int n_max,
                       void *__restrict__ outv /*[n_max][4][256] fp16 (X3: fp32)*/, const half_t *__restrict__ zero_page)
{
    __shared__ __attribute__((aligned(16))) unsigned char X[(X3 ? 2 : 1) * SD_XB];
    __shared__ int geo[2 * SD_KP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;
    int n = n_max;
    const int k0 = blockIdx.x * SD_KP;
    // the key points are requested BEFORE the count is known (two round trips to memory side by side instead of one behind the other): rows of
    // the list beyond the count hold whatever they hold -- such a row's patch addresses are bounds-checked like any other and its outputs are
    // not stored
    float kpx = 0.0f, kpy = 0.0f;
    if (tid < SD_KP) {
        const int kp = k0 + tid < n_max ? k0 + tid : n_max - 1;
        kpx = kpts[2 * kp];
        kpy = kpts[2 * kp + 1];
    }
    const int n0 = blockIdx.y * 128 + wave * 32;            // this wave's output channels
    // Filter fragments of one tap: four K slices of 16 channels, straight from the packed filters in L2.  Plain fp16: the fragments of tap
    // t + DIST (running on across the chunks) are requested BEFORE the MFMAs of tap t, the first DIST before the set-up -- as first written (load, wait, eight
    // MFMAs per tap) every tap paid an L2 round trip that only the other resident block could hide: 36 round trips per block, 84 k cycles for
    // 9.2 k cycles of MFMA issue per wave (profiles/r05v_sparse_da3_ab.txt).
    // LINA: in the layers' common packing ([chunk of 32][tap][CoutP][32]) a fragment load touches sixteen 128-byte lines and uses half of each (the
    // odd K slice the other half); the kernel is bound by what its waves pull through the CU's vector L1 -- 1.18 MB of fragments per CU and launch,
    // ~40 B/clk while the taps run (a trace with producer waves had the patch copies crawl at 3 B/clk beside them) -- so the repacked array, one
    // contiguous kilobyte per load, is worth 3.5 us
#define SD_LOAD_A(dst_, c_, tap_)                                                                                          \
    _Pragma("unroll") for (int k16 = 0; k16 < 4; ++k16)                                                                   \
        dst_[k16] = LINA ? *reinterpret_cast<const h8_t *>(wpk + ((size_t)(((c_)*2 + (k16 >> 1)) * 9 + (tap_)) * CoutP + n0) * 32 + (k16 & 1) * 512 + lane * 8) \
                         : *reinterpret_cast<const h8_t *>(wpk + ((size_t)(((c_)*2 + (k16 >> 1)) * 9 + (tap_)) * CoutP + n0 + lrow) * 32 + (k16 & 1) * 16 + lhi * 8);
    constexpr int RING = X3 ? SFD2_SD_RING_X3 : SFD2_SD_RING, DIST = RING - 1;
    h8_t an[RING][4], aln[X3 ? RING : 1][4];                  // ring: DIST taps in flight behind the one in use (X3: the hi and the lo' fragments)
    constexpr bool PF = SFD2_SD_PF && RING > 1;
    static_assert(PF || !LINA, "the repacked filters are read by the prefetching form");
    const half_t *wpl = wpk + (size_t)8 * 9 * CoutP * 32;     // X3: the lo' plane of the filters
#define SD_LOAD_AL(dst_, c_, tap_)                                                                                         \
    _Pragma("unroll") for (int k16 = 0; k16 < 4; ++k16)                                                                   \
        dst_[k16] = LINA ? *reinterpret_cast<const h8_t *>(wpl + ((size_t)(((c_)*2 + (k16 >> 1)) * 9 + (tap_)) * CoutP + n0) * 32 + (k16 & 1) * 512 + lane * 8) \
                         : *reinterpret_cast<const h8_t *>(wpl + ((size_t)(((c_)*2 + (k16 >> 1)) * 9 + (tap_)) * CoutP + n0 + lrow) * 32 + (k16 & 1) * 16 + lhi * 8);
    if (PF) {
#pragma unroll
        for (int i = 0; i < DIST; ++i) {
            SD_LOAD_A(an[i], 0, i)
            if (X3) { SD_LOAD_AL(aln[i], 0, i) }
        }
    }
    if (count) { const unsigned int c = *count; if ((unsigned int)n > c) n = (int)c; }
    if (k0 >= n) return;

    if (tid < SD_KP) {
        int x0, y0;
        if (k0 + tid >= n) kpx = kpy = 0.0f;        // a row beyond the count holds stale or uninitialised bits: no float -> int conversion of those
        sd_corner(kpx, kpy, half_w, half_h, hc, wc, x0, y0);
        geo[2 * tid] = x0;
        geo[2 * tid + 1] = y0;
    }
    __syncthreads();

    f32x16_t acc[2], acl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; acl[t][r] = 0.0f; }

    // B fragments: lane -> pixel p = lrow of tile t: key point t * 8 + (p >> 2), corner p & 3 (row = corner >> 1, column = corner & 1)
    int recb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) recb[t] = (t * 8 + (lrow >> 2)) * 16 + ((lrow >> 1) & 1) * 4 + (lrow & 1);
    const int kcode = (lrow >> 2) & 3;                     // the key point's part of the slot swizzle (header)

#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c) __syncthreads();                               // every wave is past its reads of the previous chunk
        // copies: instruction j = wave + 4 * i moves records 8 j .. 8 j + 7 (half a key point's patch), lane -> (record, slot)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = wave + 4 * i;
            const int kpl = j >> 1, ppix = (j & 1) * 8 + (lane >> 3), slot = lane & 7;
            const int x0 = geo[2 * kpl], y0 = geo[2 * kpl + 1];
            const int iy = y0 - 1 + (ppix >> 2), ix = x0 - 1 + (ppix & 3);
            const int part = slot ^ (ppix & 7) ^ (kpl & 3);
            const bool ok = iy >= 0 && iy < hc && ix >= 0 && ix < wc;
            const half_t *src = ok ? fmap + ((size_t)iy * wc + ix) * 256 + c * 64 + part * 8 : zero_page + slot * 8;
            __builtin_amdgcn_global_load_lds((sd_gbl_t *)src, (sd_lds_t *)(X + j * 1024), 16, 0, 0);
            if (X3) {
                const half_t *srl = ok ? fmap_lo + ((size_t)iy * wc + ix) * 256 + c * 64 + part * 8 : zero_page + slot * 8;
                __builtin_amdgcn_global_load_lds((sd_gbl_t *)srl, (sd_lds_t *)(X + SD_XB + j * 1024), 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            h8_t a[4], al[4];
            if (!PF) {
#pragma unroll
                for (int k16 = 0; k16 < 4; ++k16) {
                    const size_t ao = ((size_t)((c * 2 + (k16 >> 1)) * 9 + tap) * CoutP + n0 + lrow) * 32 + (k16 & 1) * 16 + lhi * 8;
                    a[k16] = *reinterpret_cast<const h8_t *>(wpk + ao);
                    if (X3) al[k16] = *reinterpret_cast<const h8_t *>(wpk + (size_t)8 * 9 * CoutP * 32 + ao);   // the lo' plane of the filters
                }
            } else {
#pragma unroll
                for (int k16 = 0; k16 < 4; ++k16) {
                    a[k16] = an[(c * 9 + tap) % RING][k16];
                    if (X3) al[k16] = aln[(c * 9 + tap) % RING][k16];
                }
                // the fragments DIST taps on (the last chunk's last taps re-read the last tap's: no branch, the values are not used)
                const int gn = c * 9 + tap + DIST < 36 ? c * 9 + tap + DIST : 35;
                SD_LOAD_A(an[(c * 9 + tap + DIST) % RING], gn / 9, gn % 9)
                if (X3) { SD_LOAD_AL(aln[(c * 9 + tap + DIST) % RING], gn / 9, gn % 9) }
                __builtin_amdgcn_sched_barrier(0);            // the requests stay in front of this tap's MFMAs
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int rec = recb[t] + ky * 4 + kx;
#pragma unroll
                for (int k16 = 0; k16 < 4; ++k16) {
                    const h8_t b = *reinterpret_cast<const h8_t *>(X + rec * 128 + (((k16 * 2 + lhi) ^ (rec & 7) ^ kcode) << 4));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k16], b, acc[t], 0, 0, 0);
                    if (X3) {
                        const h8_t bl = *reinterpret_cast<const h8_t *>(X + SD_XB + rec * 128 + (((k16 * 2 + lhi) ^ (rec & 7) ^ kcode) << 4));
                        acl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[k16], b, acl[t], 0, 0, 0);
                        acl[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k16], bl, acl[t], 0, 0, 0);
                    }
                }
            }
        }
    }
#undef SD_LOAD_A
#undef SD_LOAD_AL

    // C layout: lane owns pixel (t * 32 + lrow), channels n0 + 8 q + 4 lhi + j
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int kp = k0 + t * 8 + (lrow >> 2);
        if (kp >= n) continue;
        const size_t oo = ((size_t)kp * 4 + (lrow & 3)) * 256 + n0 + 4 * lhi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sc = *reinterpret_cast<const float4 *>(scale + n0 + 8 * q + 4 * lhi);
            const float4 sh = *reinterpret_cast<const float4 *>(shift + n0 + 8 * q + 4 * lhi);
            float a0 = acc[t][4 * q + 0], a1 = acc[t][4 * q + 1], a2 = acc[t][4 * q + 2], a3 = acc[t][4 * q + 3];
            if (X3) {   // (conv_igemm_x3_kernel's combination of the two accumulators)
                a0 += acl[t][4 * q + 0] * (1.0f / 2048.0f); a1 += acl[t][4 * q + 1] * (1.0f / 2048.0f);
                a2 += acl[t][4 * q + 2] * (1.0f / 2048.0f); a3 += acl[t][4 * q + 3] * (1.0f / 2048.0f);
            }
            float v0 = a0 * sc.x + sh.x, v1 = a1 * sc.y + sh.y, v2 = a2 * sc.z + sh.z, v3 = a3 * sc.w + sh.w;
            if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
            if (X3) {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + oo + 8 * q) = make_float4(v0, v1, v2, v3);
            } else {
                h4_t hv = {(half_t)v0, (half_t)v1, (half_t)v2, (half_t)v3};
                *reinterpret_cast<h4_t *>(reinterpret_cast<half_t *>(outv) + oo + 8 * q) = hv;
            }
        }
    }
}

// [chunk of 32][tap][CoutP][32] -> [chunk][tap][CoutP / 32][K half][lane = lhi * 32 + row][8]: element (row r of group g, K half h, lane half lhi, e)
// comes from channel h * 16 + lhi * 8 + e of filter g * 32 + r
__global__ void sparse_da3_repack_kernel(const half_t *__restrict__ src, half_t *__restrict__ dst, int CoutP, int n_ct /* chunks x taps */)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece
    if (i >= (size_t)n_ct * CoutP * 4) return;
    const int lane = (int)(i & 63), h = (int)((i >> 6) & 1);
    const size_t grp = i >> 7;                                           // (chunk-tap) * (CoutP / 32) + group
    const size_t ct = grp / (CoutP / 32), g = grp % (CoutP / 32);
    const int r = lane & 31, lhi = lane >> 5;
    *reinterpret_cast<h8_t *>(dst + i * 8) = *reinterpret_cast<const h8_t *>(src + (ct * CoutP + g * 32 + r) * 32 + h * 16 + lhi * 8);
}

void launch_sparse_da3_repack(hipStream_t st, const half_t *w, half_t *dst, int CoutP, int cin)
{
    const int n_ct = cin / 32 * 9;
    const size_t pieces = (size_t)n_ct * CoutP * 4;
    hipLaunchKernelGGL(sparse_da3_repack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, w, dst, CoutP, n_ct);
}

// wsl: the repacked filters (ConvW::wsl); -DSFD2_EXPERIMENTS builds: SFD2_SD_LINA=0 reads the common packing (the A side of the A/B)
void launch_sparse_da3(hipStream_t st, const half_t *fmap, int hc, int wc, int nh, int nw, const half_t *wpk, const half_t *wsl, int CoutP,
                       const float *scale, const float *shift, int relu, const float *kpts, const unsigned int *count, int n_max,
                       half_t *out, const half_t *zero_page)
{
    if (n_max <= 0) return;
    static const char *force = sfd2_env("SFD2_SD_LINA");
    if (wsl && !(force && force[0] == '0'))
        hipLaunchKernelGGL((sparse_da3_kernel<false, true>), dim3((n_max + SD_KP - 1) / SD_KP, 2), dim3(SD_NT), 0, st, fmap, nullptr, hc, wc, (float)nw / 2.0f,
                           (float)nh / 2.0f, wsl, CoutP, scale, shift, relu, kpts, count, n_max, out, zero_page);
    else
        hipLaunchKernelGGL((sparse_da3_kernel<false, false>), dim3((n_max + SD_KP - 1) / SD_KP, 2), dim3(SD_NT), 0, st, fmap, nullptr, hc, wc, (float)nw / 2.0f,
                           (float)nh / 2.0f, wpk, CoutP, scale, shift, relu, kpts, count, n_max, out, zero_page);
}

// SFD2_PREC_F16X3: planes in, [hi][lo'] filters (wsl: both planes repacked, or null), fp32 out
void launch_sparse_da3_x3(hipStream_t st, const half_t *fmap_hi, const half_t *fmap_lo, int hc, int wc, int nh, int nw, const half_t *wpk, const half_t *wsl,
                          int CoutP, const float *scale, const float *shift, int relu, const float *kpts, const unsigned int *count,
                          int n_max, float *out, const half_t *zero_page)
{
    if (n_max <= 0) return;
    static const char *force = sfd2_env("SFD2_SD_LINA");
#if SFD2_SD_PF && SFD2_SD_RING_X3 > 1
    if (wsl && !(force && force[0] == '0')) {
        hipLaunchKernelGGL((sparse_da3_kernel<true, true>), dim3((n_max + SD_KP - 1) / SD_KP, 2), dim3(SD_NT), 0, st, fmap_hi, fmap_lo, hc, wc,
                           (float)nw / 2.0f, (float)nh / 2.0f, wsl, CoutP, scale, shift, relu, kpts, count, n_max, out, zero_page);
        return;
    }
#endif
    (void)force;
    hipLaunchKernelGGL((sparse_da3_kernel<true, false>), dim3((n_max + SD_KP - 1) / SD_KP, 2), dim3(SD_NT), 0, st, fmap_hi, fmap_lo, hc, wc,
                       (float)nw / 2.0f, (float)nh / 2.0f, wpk, CoutP, scale, shift, relu, kpts, count, n_max, out, zero_page);
}
