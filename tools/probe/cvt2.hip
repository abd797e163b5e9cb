#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, unsigned *out, float scale)
{
    int i = threadIdx.x;
    v2s old = {0, 0};
    v2s r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[2 * i], in[2 * i + 1], scale, false);
    unsigned u; __builtin_memcpy(&u, &r, 4);
    out[i] = u;
    v2s r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[2 * i], in[2 * i + 1], scale, true);
    __builtin_memcpy(&u, &r2, 4);
    out[64 + i] = u;
    // clamp modifier on the plain conversion (inline asm)
    unsigned c;
    c = 0;
    out[128 + i] = c;
}
static float e4m3_to_f(unsigned char v)
{
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) x = NAN;
    else x = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main()
{
    float vals[16] = {1.f, 3.3f, 0.01f, 500.f, 448.f, 1000.f, 0.001953125f, 100000.f, -2.5f, -600.f, 7.9f, 0.3f, 2000.f, 1800.f, 464.f, 480.f};
    float *d; unsigned *o; hipMalloc(&d, 64 * 2 * 4); hipMalloc(&o, 192 * 4);
    hipMemset(d, 0, 512); hipMemcpy(d, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (float scale : {1.0f, 4.0f, 0.25f, 0.001953125f}) {
        k<<<1, 64>>>(d, o, scale);
        unsigned h[192]; hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
        printf("scale %g:\n", scale);
        for (int i = 0; i < 8; ++i)
            printf("  (%g, %g) -> lo-word %04x = (%g, %g) | hi-word sel: %08x | clamp-asm %04x = (%g, %g)\n", vals[2 * i], vals[2 * i + 1], h[i] & 0xffff,
                   e4m3_to_f(h[i] & 255), e4m3_to_f((h[i] >> 8) & 255), h[64 + i], h[128 + i] & 0xffff, e4m3_to_f(h[128 + i] & 255), e4m3_to_f((h[128 + i] >> 8) & 255));
    }
    return 0;
}
