// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with A = fp6 (e2m3, cbsz = 2) and B = fp8 (e4m3): where does K index k of
// row i sit in the lane's registers, what does a 6-bit code mean, and whose scale byte applies to which (row, K block)?
// B[k][j] = f(k) for every column j, f(k) = (8 + (k & 7)) * 2^((k >> 3) - 7): 64 distinct e4m3 values, so C[i][j] = sum_k A[i][k] f(k)
// names the k of a single non-zero A entry.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ unsigned char e4m3_of(int k) { const int m = k & 7, e = (k >> 3) - 7 + 3 + 7; return (unsigned char)((e << 3) | m); }   // (8 + m) * 2^((k>>3) - 7)
__global__ void probe(float *out, int slot, int code, int lane_set, int scale_lane, int scale_byte)
{
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0; b[i] = 0; }
    // B: lane (col = lane & 31, khalf = lane >> 5) holds K = 32 khalf .. + 31, one byte each
    for (int kk = 0; kk < 32; ++kk) {
        const int k = (lane >> 5) * 32 + kk;
        b[kk >> 2] |= (int)e4m3_of(k) << ((kk & 3) * 8);
    }
    // A: in lane `lane_set` put `code` into 6-bit slot `slot` (bits 6 slot .. 6 slot + 5 of the 192-bit little-endian string)
    if (lane == lane_set) {
        const int bit = slot * 6;
        unsigned long long lo = (unsigned long long)code << (bit & 31);
        a[bit >> 5] |= (int)(unsigned)lo;
        if ((bit >> 5) + 1 < 8) a[(bit >> 5) + 1] |= (int)(unsigned)(lo >> 32);
    }
    int sa = 0x7f7f7f7f;
    if (lane == scale_lane) sa = (scale_byte & 255) * 0x01010101;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 0, 0, sa, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
int main()
{
    float *d; hipMalloc(&d, 64 * 16 * 4);
    float h[64 * 16];
    auto run = [&](int slot, int code, int lane_set, int scale_lane, int scale_byte) {
        probe<<<1, 64>>>(d, slot, code, lane_set, scale_lane, scale_byte);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        // C layout: lane (col = lane & 31, half = lane >> 5), register r -> row (r & 3) + 8 (r >> 2) + 4 half
        int nz = 0; float v = 0; int row = -1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (h[l * 16 + r] != 0.f) { if (!nz) { v = h[l * 16 + r]; row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); } ++nz; }
        int k = -1; float val = 0;
        for (int kk = 0; kk < 64 && k < 0; ++kk) { const float f = (8 + (kk & 7)) * ldexpf(1.f, (kk >> 3) - 7); const float q = v / f;
            for (int c6 = 1; c6 < 64; ++c6) { const int e = (c6 >> 3) & 3, m = c6 & 7; float x = e ? (8 + m) * ldexpf(1.f, e - 4) : m * 0.125f; if (c6 & 32) x = -x; if (x == q && c6 == code) { k = kk; val = x; break; } } }
        printf("slot %2d code %2d lane %2d scale(lane %2d)=%3d: %d non-zero outputs, row %d, value %g -> K index %d, code value %g\n", slot, code, lane_set, scale_lane, scale_byte, nz, row, v, k, val);
    };
    for (int s : {0, 1, 5, 6, 17, 31}) run(s, 8, 0, -1, 127);      // code 8 = 1.0 ?
    run(3, 8, 32, -1, 127); run(3, 8, 37, -1, 127);                 // upper half-wave: K 32 .. 63 of row lane - 32 ?
    for (int c6 : {1, 7, 9, 15, 16, 24, 31, 40}) run(2, c6, 4, -1, 127);
    run(2, 8, 4, 4, 128); run(2, 8, 4, 36, 128); run(2, 8, 36, 36, 129); run(2, 8, 36, 4, 129);   // whose scale byte?
    return 0;
}
