// SFD2_PREC_F16C, option "rb_inner" = 2: ResBlock.conv2 (3x3, groups = 32) + BN + ReLU and ResBlock.conv3 (1x1) + BN + residual +
// ReLU in ONE kernel (nets/sfd2.py:14-18, :32-55).  With t1 and t2 plain fp16 the grouped conv's output tile of 4 x 32 pixels x 256
// channels is 64 KB: it stays in LDS, in the record layout conv1x1_c256_c_kernel stages its input in, and never travels to
// HBM (2 x 61 MB per block at 1600x1200) -- and the block is one launch shorter.
//
//   phase G  gconv_c_kernel<false, false>'s arithmetic: 64-channel chunks of the 6 x 34 patch of t1 staged through registers
//            into LDS (double buffered, one barrier per chunk), two groups per 16x16x32 MFMA with block-diagonal filter
//            fragments, second fp16 pass with the filter residuals; a wave = one group pair of the chunk x two tile rows
//   phase C  conv1x1_c256_c_kernel<true, false, true>'s arithmetic on the tile's four 32-pixel rows: a wave owns 32 output
//            channels, filters (fp16 + fp16 residuals, fragment order) in 128 registers for the life of the block
//
// Persistent, one block per CU; the same operations in the same order as the two kernels it replaces: results are identical
// bit for bit (tests/test_gpu_f16c.py).
#include "sfd2_internal.h"

#define R23_NT 512
#define R23_TH 4
#define R23_TW 32
#define R23_PH 6
#define R23_PW 34
#define R23_NPIX (R23_PH * R23_PW)
#define R23_GCP 72                                  // halves per patch record: 64 channels + 8 of padding
#define R23_XB (R23_NPIX * R23_GCP * 2)             // bytes of one patch buffer
#define R23_T2B (R23_TH * R23_TW * 512)
#define R23_NLD ((R23_NPIX * 8 + R23_NT - 1) / R23_NT)

typedef float r23_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int r23_xcd_swizzle(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__global__ __launch_bounds__(R23_NT, 2)
void rb23_c_kernel(const half_t *__restrict__ t1, int H, int W,
                   const half_t *__restrict__ w2h /*[16 pairs][5 steps][64 lanes][8]*/, const half_t *__restrict__ w2l /*residuals * 2^11, same layout*/,
                   const float *__restrict__ sc2, const float *__restrict__ sh2,
                   const half_t *__restrict__ w3h /*fragment order [8 waves][8][64 lanes][16]*/, const half_t *__restrict__ w3l,
                   const float *__restrict__ sc3, const float *__restrict__ sh3,
                   const half_t *__restrict__ res, const half_t *__restrict__ res_c,
                   half_t *__restrict__ out, half_t *__restrict__ out_c, int tiles_x, int n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *T2 = smem;                                              // [128 pixels][512 B], 16-byte slots XOR (pixel & 31)
    half_t *XP = reinterpret_cast<half_t *>(smem + R23_T2B);               // [2][R23_NPIX][R23_GCP]
    float *SS = reinterpret_cast<float *>(smem + R23_T2B + 2 * R23_XB);    // sc2, sh2, sc3, sh3: 256 floats each

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhi = lane >> 5;            // phase C
    const int g = lane >> 4, lcol = lane & 15;              // phase G
    const int pw = wave & 3, rh = wave >> 2;

    // conv3's filters: resident
    h8_t ah[16];
    v8i_t al[8];
    {
        const size_t fo = ((size_t)wave * 8 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[2 * c] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16);
            ah[2 * c + 1] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16 + 8);
            al[c] = sfd2_cat8(*reinterpret_cast<const h8_t *>(w3l + fo + (size_t)c * 64 * 16), *reinterpret_cast<const h8_t *>(w3l + fo + (size_t)c * 64 * 16 + 8));
        }
    }
    for (int t = tid; t < 256; t += R23_NT) { SS[t] = sc2[t]; SS[256 + t] = sh2[t]; SS[512 + t] = sc3[t]; SS[768 + t] = sh3[t]; }

    // a patch chunk = 204 pixels x 8 parts of 16 bytes; piece p = tid + k * 512: pixel p >> 3 (row = pixel / 34 by multiplication)
    uint4 pre[R23_NLD];
#define R23_FETCH(oy0_, ox0_, chunk_)                                                                     \
    _Pragma("unroll") for (int k = 0; k < R23_NLD; ++k) {                                                 \
        int p = tid + k * R23_NT;                                                                         \
        asm volatile("" : "+v"(p));   /* (not hoisted out of the tile loop: 12 registers) */               \
        const int q = p >> 3;                                                                             \
        const int py = (q * 241) >> 13, px = q - py * R23_PW;                                             \
        const int iy = (oy0_)-1 + py, ix = (ox0_)-1 + px;                                                  \
        uint4 v = make_uint4(0, 0, 0, 0);                                                                 \
        if (p < R23_NPIX * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)                                   \
            v = *reinterpret_cast<const uint4 *>(t1 + (size_t)(iy * W + ix) * 256 + (chunk_)*64 + (p & 7) * 8); \
        pre[k] = v;                                                                                       \
    }

    int tile = blockIdx.x;
    int oy0, ox0;
    {
        const int swz = r23_xcd_swizzle(tile, n_tiles);
        oy0 = (swz / tiles_x) * R23_TH; ox0 = (swz % tiles_x) * R23_TW;
    }
    R23_FETCH(oy0, ox0, 0)

    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < n_tiles;
        int noy0 = 0, nox0 = 0;
        if (has_next) {
            const int swz = r23_xcd_swizzle(next, n_tiles);
            noy0 = (swz / tiles_x) * R23_TH; nox0 = (swz % tiles_x) * R23_TW;
        }
        // ------------------------------------------------ phase G: grouped 3x3 -> T2
#pragma unroll 1
        for (int chunk = 0; chunk < 4; ++chunk) {
            half_t *Xb = XP + (chunk & 1) * (R23_NPIX * R23_GCP);
#pragma unroll
            for (int k = 0; k < R23_NLD; ++k) {
                int p = tid + k * R23_NT;
                asm volatile("" : "+v"(p));
                if (p < R23_NPIX * 8) *reinterpret_cast<uint4 *>(Xb + (p >> 3) * R23_GCP + (p & 7) * 8) = pre[k];
            }
            const int pair = chunk * 4 + pw;
            h8_t wh[5], wl[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                wh[s] = *reinterpret_cast<const h8_t *>(w2h + ((size_t)(pair * 5 + s) * 64 + lane) * 8);
                wl[s] = *reinterpret_cast<const h8_t *>(w2l + ((size_t)(pair * 5 + s) * 64 + lane) * 8);
            }
            // one barrier per chunk: the buffer just written was last read two chunks ago, and every wave has passed the
            // barrier in between (T2: last read in the previous tile's phase C, which every wave left before this barrier)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (chunk < 3) { R23_FETCH(oy0, ox0, chunk + 1) }
            else if (has_next) { R23_FETCH(noy0, nox0, 0) }

            int lc = lcol;
            asm volatile("" : "+v"(lc));   // the 20 fragment addresses are recomputed per chunk (hoisted they are 20 registers)
            const int c0 = pair * 16 + g * 4;
            const float4 sc = sfd2_lds_f4(SS + c0);
            const float4 sh = sfd2_lds_f4(SS + 256 + c0);
#pragma unroll
            for (int th = 0; th < 2; ++th) {               // one tile row at a time: half the accumulators live
                const int row = rh * 2 + th;
                r23_f4 acc[2], acl[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) { acc[t] = (r23_f4){0.0f, 0.0f, 0.0f, 0.0f}; acl[t] = acc[t]; }
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    int tap = 2 * s + (g >> 1);
                    if (tap > 8) tap = 8;                   // zero-weight slot: read any valid location
                    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int q = (row + ky) * R23_PW + t * 16 + lc + kx;
                        const h8_t bh = *reinterpret_cast<const h8_t *>(Xb + q * R23_GCP + pw * 16 + (g & 1) * 8);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh, acc[t], 0, 0, 0);
                        acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh, acl[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], 1.0f / 2048.0f, acc[t][r]);
                    uint2 hv, cv;
                    sfd2_epi4<false>(acc[t][0], acc[t][1], acc[t][2], acc[t][3], sc, sh, sc, 0.0f, hv, cv);
                    const int pl = row * 32 + t * 16 + lc;
                    const int slot = (pair * 2 + (g >> 1)) ^ (pl & 31);
                    *reinterpret_cast<uint2 *>(T2 + pl * 512 + (slot << 4) + (g & 1) * 8) = hv;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // T2 complete
        // ------------------------------------------------ phase C: 1x1 + residual over the tile's four rows
#pragma unroll 1
        for (int r4 = 0; r4 < R23_TH; ++r4) {
            const int oy = oy0 + r4, ox = ox0 + lrow;
            const bool inb = oy < H && ox < W;
            const size_t obase = (size_t)(inb ? oy * W + ox : 0) * 256 + wave * 32;
            uint4 rq[2], rc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                rq[m] = *reinterpret_cast<const uint4 *>(res + obase + 8 * (2 * m + lhi));
                rc[m] = *reinterpret_cast<const uint4 *>(res_c + obase + 8 * (2 * m + lhi));
            }
            f32x16_t acc, acl;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acl[r] = 0.0f; }
            int sw = lrow;
            asm volatile("" : "+v"(sw));   // (the 16 fragment addresses: per row, not hoisted)
            const unsigned char *xp = T2 + (r4 * 32 + sw) * 512;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], b, acc, 0, 0, 0);
                acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(al[kk >> 1], kk & 1), b, acl, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(acl[r], 1.0f / 2048.0f, acc[r]);

            const int cl = wave * 32 + 4 * lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(rq[m].x, rq[m].z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(rq[m].y, rq[m].w, false, false);
                const uint2 rp[2] = {make_uint2(s0[0], s1[0]), make_uint2(s0[1], s1[1])};
                const auto c0 = __builtin_amdgcn_permlane32_swap(rc[m].x, rc[m].z, false, false);
                const auto c1 = __builtin_amdgcn_permlane32_swap(rc[m].y, rc[m].w, false, false);
                const uint2 rcp[2] = {make_uint2(c0[0], c1[0]), make_uint2(c0[1], c1[1])};
                uint2 pk[2], ck[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const float4 sc = sfd2_lds_f4(SS + 512 + cl + 8 * q);
                    const float4 sh = sfd2_lds_f4(SS + 768 + cl + 8 * q);
                    h4_t rr;
                    __builtin_memcpy(&rr, &rp[j], 8);
                    const float4 ad = make_float4((float)rr[0] + sfd2_corr_lo(rcp[j].x, 0), (float)rr[1] + sfd2_corr_lo(rcp[j].x, 1),
                                                  (float)rr[2] + sfd2_corr_lo(rcp[j].y, 0), (float)rr[3] + sfd2_corr_lo(rcp[j].y, 1));
                    sfd2_epi4<true>(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], sc, sh, ad, 0.0f, pk[j], ck[j]);
                }
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1v = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                if (inb) {
                    *reinterpret_cast<uint4 *>(out + obase + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1v[0], t0[1], t1v[1]);
                    *reinterpret_cast<uint4 *>(out_c + obase + 8 * (2 * m + lhi)) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                }
            }
        }
        if (!has_next) break;
        tile = next; oy0 = noy0; ox0 = nox0;
    }
#undef R23_FETCH
}

// t1: ResBlock.conv1's output, plain fp16 [H][W][256]; res / res_c: the block's input (hi + corr planes); out / out_c: its output
void launch_rb23_c(hipStream_t st, const half_t *t1, int H, int W, const half_t *w2h, const half_t *w2l, const float *sc2,
                   const float *sh2, const half_t *w3h, const half_t *w3l, const float *sc3, const float *sh3,
                   const half_t *res, const half_t *res_c, half_t *out, half_t *out_c)
{
    constexpr size_t lds = (size_t)R23_T2B + 2 * R23_XB + 4 * 256 * sizeof(float);
    static bool attr_done = false;
    static int slots = 256;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(rb23_c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int tiles_x = (W + R23_TW - 1) / R23_TW, tiles_y = (H + R23_TH - 1) / R23_TH;
    const int n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return;
    const int grid = n_tiles < slots ? n_tiles : slots;
    hipLaunchKernelGGL(rb23_c_kernel, dim3(grid), dim3(R23_NT), lds, st, t1, H, W, w2h, w2l, sc2, sh2, w3h, w3l, sc3, sh3, res, res_c,
                       out, out_c, tiles_x, n_tiles);
}
