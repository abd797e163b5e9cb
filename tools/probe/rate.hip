#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters, int seed)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    v8i a8, b8; h8 ah[2], bh[2];
    unsigned x = threadIdx.x * 2654435761u + seed;
    for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; a8[i] = (x & 0x7f7f7f7f) % 0x78787878; x = x * 1664525u + 1013904223u; b8[i] = (x & 0x77777777); }
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; ah[k][i] = (_Float16)((float)(x >> 16) / 65536.f - 0.5f); x = x * 1664525u + 1013904223u; bh[k][i] = (_Float16)((float)(x >> 16) / 65536.f - 0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (MODE == 0 || MODE == 2) {
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], acc[a], 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2)
                acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[a], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 2048;
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) rate<0><<<grid, 256>>>(d, iters, rep); else if (mode == 1) rate<1><<<grid, 256>>>(d, iters, rep); else rate<2><<<grid, 256>>>(d, iters, rep);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // "chunk" = 32 channels of main (2 f16 MFMAs, 65536 flop) or 1 fp8 MFMA (131072 flop)
            double chunks = (double)grid * 4 * iters * 4;
            printf("mode %d (%s) rep %d: %.3f ms, %.1f ns per wave-chunk-step, f16-equiv %.0f TF/s, raw %.0f TF/s\n", mode, mode == 0 ? "2x f16 32x32x16" : mode == 1 ? "1x fp8 32x32x64" : "both", rep, ms,
                   ms * 1e6 / (iters * 4.0), chunks * 65536 * (mode == 2 ? 2 : 1) / (ms * 1e-3) / 1e12, chunks * (mode == 0 ? 65536.0 : mode == 1 ? 131072.0 : 196608.0) / (ms * 1e-3) / 1e12);
        }
    return 0;
}
