// Probe: issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 by operand format (two waves per SIMD, four accumulator chains per wave,
// random operand bits): cbsz / blgp = 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1.  K = 64 per instruction in every format.
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f8f6f4.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FA, int FB>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, int seed)
{
    f32x16 c[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) c[a][r] = 0.f;
    v8i A[4], B[4];
    unsigned x = threadIdx.x * 2654435761u + seed;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; A[a][i] = x & 0x37373737; x = x * 1664525u + 1013904223u; B[a][i] = x & 0x37373737; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int a = 0; a < 4; ++a)
                c[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[(a + rep) & 3], B[a], c[a], FA, FB, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += c[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int FA, int FB> static void run(float *d, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 512;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k<FA, FB><<<grid, 256>>>(d, iters, rep);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)grid * 4 * iters * 16;      // MFMA instructions
        printf("%-12s rep %d: %.3f ms, %.1f ns per MFMA per SIMD, %.0f TFLOP/s (K = 64)\n", name, rep, ms, ms * 1e6 / (iters * 16.0 * 2), n * 131072.0 / (ms * 1e-3) / 1e12);
    }
}
int main()
{
    float *d; hipMalloc(&d, 512 * 256 * 4);
    run<0, 0>(d, "fp8 x fp8");
    run<2, 2>(d, "fp6 x fp6");
    run<4, 4>(d, "fp4 x fp4");
    run<0, 2>(d, "fp8 x fp6");
    run<0, 4>(d, "fp8 x fp4");
    return 0;
}
