// Probe: v_cvt_scalef32_2xpk16_fp6_f32 (32 floats -> 32 e2m3 codes in 6 registers) and its inverse v_cvt_scalef32_pk32_f32_fp6: where
// do src0[j] / src1[j] land, what does the scale operand mean, how does it round and saturate?
// hipcc --offload-arch=gfx950 -O3 tools/probe/cvt_fp6.hip -o /tmp/cvt_fp6 && /tmp/cvt_fp6
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef int v6i __attribute__((ext_vector_type(6)));
__global__ void k(float *out, unsigned *raw, float scale_enc, float scale_dec, int mode)
{
    f16v a, b;
    for (int j = 0; j < 16; ++j) {
        if (mode == 0) { a[j] = 1.0f + 0.25f * j; b[j] = -(0.125f * (j + 1)); }       // distinct, exactly representable (<= 4.75; subnormals and small normals)
        if (mode == 1) { a[j] = 0.0625f * j; b[j] = 4.0f + 0.25f * j; }                  // halves of the subnormal step (rounding), values up to 7.75 (saturation)
        if (mode == 2) { a[j] = 3.0f * (j + 1); b[j] = 1.0f + 0.0625f * j; }              // large values (scale), ties between normals
    }
    v6i d;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(scale_enc));
    f32v r;
    asm volatile("v_cvt_scalef32_pk32_f32_fp6 %0, %1, %2" : "=&v"(r) : "v"(d), "v"(scale_dec));
    if (threadIdx.x == 0) {
        for (int j = 0; j < 32; ++j) out[j] = r[j];
        for (int j = 0; j < 6; ++j) raw[j] = (unsigned)d[j];
    }
}
int main()
{
    float *out; unsigned *raw;
    hipMalloc(&out, 32 * 4); hipMalloc(&raw, 6 * 4);
    float h[32]; unsigned hr[6];
    const float se[] = {1.0f, 1.0f, 1.0f, 2.0f, 8.0f, 8.0f}, sd[] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 8.0f};
    const int md[] = {0, 1, 2, 0, 2, 2};
    for (int t = 0; t < 6; ++t) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, raw, se[t], sd[t], md[t]);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hr, raw, sizeof(hr), hipMemcpyDeviceToHost);
        printf("mode %d scale_enc %g scale_dec %g\n  raw:", md[t], se[t], sd[t]);
        for (int j = 0; j < 6; ++j) printf(" %08x", hr[j]);
        printf("\n  codes (6 bits each, sequential):");
        for (int j = 0; j < 32; ++j) { const int bit = 6 * j; unsigned long long w = hr[bit >> 5] | ((unsigned long long)(bit / 32 + 1 < 6 ? hr[bit / 32 + 1] : 0) << 32); printf(" %02x", (unsigned)((w >> (bit & 31)) & 63)); }
        printf("\n  decoded:");
        for (int j = 0; j < 32; ++j) printf(" %g", h[j]);
        printf("\n");
    }
    return 0;
}
