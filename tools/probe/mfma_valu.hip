// Probe: do VALU instructions and MFMAs of one SIMD overlap?  Two waves per SIMD (two 256-thread blocks per CU).
//   mode 0: MFMA only (two dependent chains per wave, 16 MFMAs 32x32x16 f16 per step)
//   mode 1: VALU only (160 v_max3 / v_and_or per step, four independent chains)
//   mode 2: per step 16 MFMAs, then 160 VALU (tile-by-tile, as match_mutual_kernel)
//   mode 3: per step 16 x (1 MFMA + 10 VALU), finely interleaved in program order
//   mode 4: wave-specialised: the waves of block parity 0 run mode 0, those of parity 1 mode 1 (two steps' worth each)
//   mode 5: as 2 but the VALU part consumes the MFMA results (true dependence)
//   mode 6: MFMA only, four independent chains per wave;  mode 7: the same work as 64 x v_mfma_f32_16x16x32_f16 (four chains)
//   mode 8: mode 0 with all-zero operands (data-dependent power);  mode 9: MFMA only, eight chains per wave
//   mode 10: per step 16 MFMAs in FOUR chains, then 160 VALU;  mode 11: 16 x (1 MFMA + 10 VALU) with four chains
//   mode 12: mode 10 with the accumulators in AGPRs ("a" constraint)
// Every run also reports the shader clock: s_memtime (shader cycles) against s_memrealtime (100 MHz) around the loop of block 0.
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu.hip -o /tmp/mfma_valu && /tmp/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VALU10(x0, x1, x2, x3, k)                                                         \
    asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\t"         \
                 "v_max3_f32 %2, %2, %0, %1\n\tv_and_or_b32 %3, %3, %4, %5\n\t"           \
                 "v_max3_f32 %0, %0, %2, %3\n\tv_and_or_b32 %1, %1, %4, %5\n\t"           \
                 "v_max3_f32 %2, %2, %0, %1\n\tv_and_or_b32 %3, %3, %4, %5\n\t"           \
                 "v_max3_f32 %0, %0, %2, %3\n\tv_max3_f32 %1, %1, %2, %3"                 \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(keep), "s"(k))

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters, int seed, unsigned long long *clk)
{
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    h8 a[8], b[8];
    unsigned x = threadIdx.x * 2654435761u + seed;
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; a[k][i] = (_Float16)((float)(x >> 16) / 65536.f - 0.5f); b[k][i] = (_Float16)((float)(x & 0xffff) / 65536.f - 0.5f); }
    if (MODE == 8) for (int k = 0; k < 8; ++k) for (int i = 0; i < 8; ++i) { a[k][i] = (_Float16)0.f; b[k][i] = (_Float16)0.f; }
    f32x16 c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 d[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) d[q][r] = 0.f;
    float v0 = (float)x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    unsigned keep = 0xffffff80u;
    asm volatile("" : "+v"(keep));
    const int role = MODE == 4 ? (blockIdx.x & 1) : -1;
    for (int it = 0; it < iters; ++it) {
        const int code = it & 127;
        if (MODE == 6) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 1], b[k + 1], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k + 1], a[k + 1], c3, 0, 0, 0);
            }
        }
        if (MODE == 9) {
#pragma unroll
            for (int k = 0; k < 8; k += 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 1], b[k + 1], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k + 1], a[k + 1], c3, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 2], b[k + 2], c4, 0, 0, 0);
                c5 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k + 2], a[k + 2], c5, 0, 0, 0);
                c6 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 3], b[k + 3], c6, 0, 0, 0);
                c7 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k + 3], a[k + 3], c7, 0, 0, 0);
            }
        }
        if (MODE == 10 || MODE == 11) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                if (MODE == 11) { __builtin_amdgcn_sched_barrier(0); VALU10(v0, v1, v2, v3, code); __builtin_amdgcn_sched_barrier(0); }
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
                if (MODE == 11) { __builtin_amdgcn_sched_barrier(0); VALU10(v0, v1, v2, v3, code); __builtin_amdgcn_sched_barrier(0); }
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k + 1], b[k + 1], c2, 0, 0, 0);
                if (MODE == 11) { __builtin_amdgcn_sched_barrier(0); VALU10(v0, v1, v2, v3, code); __builtin_amdgcn_sched_barrier(0); }
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k + 1], a[k + 1], c3, 0, 0, 0);
                if (MODE == 11) { __builtin_amdgcn_sched_barrier(0); VALU10(v0, v1, v2, v3, code); __builtin_amdgcn_sched_barrier(0); }
            }
            if (MODE == 10) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; ++k) VALU10(v0, v1, v2, v3, code);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 12) {
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a[k]), "v"(b[k]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(b[k]), "v"(a[k]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(a[k + 1]), "v"(b[k + 1]));
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(b[k + 1]), "v"(a[k + 1]));
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) VALU10(v0, v1, v2, v3, code);
        }
        if (MODE == 7) {
#pragma unroll
            for (int k = 0; k < 64; ++k) d[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[k & 7], b[(k >> 2) & 7], d[k & 3], 0, 0, 0);
        }
        if (MODE == 0 || MODE == 8 || role == 0) {
            for (int rep = 0; rep < (MODE == 4 ? 2 : 1); ++rep)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
                }
        }
        if (MODE == 1 || role == 1) {
            for (int rep = 0; rep < (MODE == 4 ? 2 : 1); ++rep)
#pragma unroll
                for (int k = 0; k < 16; ++k) VALU10(v0, v1, v2, v3, code);
        }
        if (MODE == 2 || MODE == 5) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], MODE == 5 && k == 0 ? (c0 * 0.f) : c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
            }
            if (MODE == 5) { v0 += c0[0]; asm volatile("" : "+v"(v0)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k) VALU10(v0, v1, v2, v3, code);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k], b[k], c0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU10(v0, v1, v2, v3, code);
                __builtin_amdgcn_sched_barrier(0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[k], a[k], c1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                VALU10(v0, v1, v2, v3, code);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) s += d[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - w0; }
}

int main()
{
    float *d; hipMalloc(&d, 512 * 256 * 4);
    unsigned long long *clk; hipMalloc(&clk, 16); unsigned long long hclk[2];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, grid = 512;
    for (int mode = 0; mode < 13; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            switch (mode) {
            case 0: probe<0><<<grid, 256>>>(d, iters, rep, clk); break;
            case 1: probe<1><<<grid, 256>>>(d, iters, rep, clk); break;
            case 2: probe<2><<<grid, 256>>>(d, iters, rep, clk); break;
            case 3: probe<3><<<grid, 256>>>(d, iters, rep, clk); break;
            case 4: probe<4><<<grid, 256>>>(d, iters, rep, clk); break;
            case 5: probe<5><<<grid, 256>>>(d, iters, rep, clk); break;
            case 6: probe<6><<<grid, 256>>>(d, iters, rep, clk); break;
            case 7: probe<7><<<grid, 256>>>(d, iters, rep, clk); break;
            case 8: probe<8><<<grid, 256>>>(d, iters, rep, clk); break;
            case 9: probe<9><<<grid, 256>>>(d, iters, rep, clk); break;
            case 10: probe<10><<<grid, 256>>>(d, iters, rep, clk); break;
            case 11: probe<11><<<grid, 256>>>(d, iters, rep, clk); break;
            default: probe<12><<<grid, 256>>>(d, iters, rep, clk); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: two waves x iters steps; a step = 16 MFMAs (512 pipe cycles) and / or 160 VALU (640 issue cycles)
            const bool has_mfma = mode != 1;
            printf("mode %d rep %d: %.3f ms = %.0f ns per step per wave (2 waves per SIMD)", mode, rep, ms, ms * 1e6 / iters);
            if (has_mfma) printf(", MFMA %.0f TFLOP/s", (double)grid * 4 * iters * 16 * 32768.0 / (ms * 1e-3) / 1e12);
            hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
            printf(", shader clock %.0f MHz\n", (double)hclk[0] / (double)hclk[1] * 100.0);
        }
    return 0;
}
