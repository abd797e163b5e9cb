#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void cvt_probe(const float *in, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        int r = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], -in[i], 0, false);
        out[i] = (unsigned)r;
    }
}

// A: [64 lanes][32 bytes], B same, scale bytes sa, sb (E8M0) uniform
__global__ void mfma_probe(const v8i *A, const v8i *B, float *C, int sa, int sb)
{
    int l = threadIdx.x;
    v8i a = A[l], b = B[l];
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) C[l * 16 + r] = c[r];
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
    float vals[] = {0.f, 1.f, 1.0625f, 1.1f, 0.001953125f, 0.0009765625f, 0.0005f, 447.f, 448.f, 460.f, 464.f, 480.f, 500.f, 1e5f, 3.3f, 0.017f};
    int n = sizeof(vals) / 4;
    float *din; unsigned *dout;
    hipMalloc(&din, 256); hipMalloc(&dout, 256);
    hipMemcpy(din, vals, n * 4, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(din, dout, n);
    unsigned ho[64];
    hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt %g -> %02x (%g)  neg %02x (%g)\n", vals[i], ho[i] & 255, e4m3_to_f(ho[i] & 255), (ho[i] >> 8) & 255, e4m3_to_f((ho[i] >> 8) & 255));
    // mfma
    unsigned char hA[64 * 32], hB[64 * 32];
    srand(1);
    for (int i = 0; i < 64 * 32; ++i) { 
        do { hA[i] = rand() & 255; } while ((hA[i] & 0x7f) == 0x7f);
        do { hB[i] = rand() & 255; } while ((hB[i] & 0x7f) == 0x7f);
    }
    v8i *dA, *dB; float *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 64 * 16 * 4);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    for (int trial = 0; trial < 3; ++trial) {
        int sa = trial == 0 ? 127 : (trial == 1 ? 124 : 127), sb = trial == 2 ? 130 : 127;
        int sav = sa * 0x01010101, sbv = sb * 0x01010101;
        mfma_probe<<<1, 64>>>(dA, dB, dC, sav, sbv);
        float hC[64 * 16];
        hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int h = 0; h < 2; ++h)
                    for (int b = 0; b < 32; ++b)
                        ref += (double)e4m3_to_f(hA[(row + 32 * h) * 32 + b]) * (double)e4m3_to_f(hB[(col + 32 * h) * 32 + b]);
                ref *= ldexp(1.0, (sa - 127) + (sb - 127));
                maxerr = fmax(maxerr, fabs(ref - hC[l * 16 + r])); maxref = fmax(maxref, fabs(ref));
            }
        printf("mfma trial %d sa %d sb %d: max err %g (max ref %g)\n", trial, sa, sb, maxerr, maxref);
    }
    return 0;
}
