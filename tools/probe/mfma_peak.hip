// Probe: what does the MI355X sustain on v_mfma_f32_32x32x16_f16 ALONE, and what is that figure a statement about?
// (the round-3 probe -- two waves per SIMD, two / four / eight chains, uniform +-0.5 operands -- read 1.11 PFLOP/s,
//  and three kernels of this repository beat it; a kernel with LDS reads and barriers cannot out-run a register-only MFMA loop on a
//  true limit, so that probe was limited by something else.)
// Matrix: waves per SIMD {1, 2, 4} x independent accumulator chains per wave {2, 4, 8} x accumulators in VGPRs / AGPRs x operand data
//   zeros        all-zero A and B (no switching in the multipliers: the power floor)
//   relu         post-ReLU-like activations: half the entries exactly zero, the rest uniform in (0, 4]; filters uniform +-0.5
//   uniform      both operands uniform +-0.5 (the round-3 probe's data: every mantissa bit toggles)
// Each cell: wall time of the kernel (HIP events), MFMAs per SIMD, ns and SHADER CYCLES per MFMA and SIMD (s_memtime around block 0's
// loop against s_memrealtime), PFLOP/s.  The shader clock under rocprofv3 (GRBM_GUI_ACTIVE / duration) comes from
// `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -- /tmp/mfma_peak` (tools/collect_round.sh writes both).
// Occupancy is pinned by LDS: a block of 256 threads = one wave per SIMD, and the launch requests 160 KB / W of dynamic LDS so that exactly
// W blocks fit a CU; the grid is W x CUs blocks.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MF_V(c_, a_, b_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c_) : "v"(a_), "v"(b_))
#define MF_A(c_, a_, b_) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c_) : "v"(a_), "v"(b_))

template <int CHAINS, bool AGPR>
__global__ __launch_bounds__(256) void peak(float *out, int iters, int data, unsigned long long *clk)
{
    extern __shared__ unsigned char pad[];
    f32x16 c[CHAINS];
#pragma unroll
    for (int q = 0; q < CHAINS; ++q)
        for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
    h8 a[4], b[4];
    unsigned x = (threadIdx.x + 256u * blockIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            const float u = (float)(x >> 16) / 65536.f, v = (float)(x & 0xffff) / 65536.f;
            float av = u - 0.5f, bv = v - 0.5f;                    // uniform
            if (data == 0) { av = 0.f; bv = 0.f; }                 // zeros
            if (data == 1) { av = (x & 0x100) ? 0.f : 4.f * u; }   // relu-like activations against +-0.5 filters
            a[k][i] = (_Float16)av;
            b[k][i] = (_Float16)bv;
        }
    if (threadIdx.x == 9999) pad[0] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // 16 MFMAs per iteration and wave; chain q is touched every CHAINS-th instruction
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (AGPR) MF_A(c[m % CHAINS], a[m & 3], b[(m >> 2) & 3]);
            else MF_V(c[m % CHAINS], a[m & 3], b[(m >> 2) & 3]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < CHAINS; ++q)
        for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

// Second question: how much VALU work rides along for free?  VALU independent v_max3_f32 / v_and_or_b32 (four register chains, no
// dependence on the accumulators) behind every MFMA, W waves per SIMD, four chains in VGPRs, relu-like operands.
#define VOP(x0_, x1_, x2_, x3_) asm volatile("v_max3_f32 %0, %0, %1, %2\n\tv_and_or_b32 %1, %1, %4, %3" : "+v"(x0_), "+v"(x1_), "+v"(x2_), "+v"(x3_) : "v"(keep))
template <int NV /* VALU instructions per MFMA, even */>
__global__ __launch_bounds__(256) void mixed(float *out, int iters, unsigned long long *clk)
{
    extern __shared__ unsigned char pad[];
    f32x16 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
    h8 a[4], b[4];
    unsigned x = (threadIdx.x + 256u * blockIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u;
            const float u = (float)(x >> 16) / 65536.f, v = (float)(x & 0xffff) / 65536.f;
            a[k][i] = (_Float16)((x & 0x100) ? 0.f : 4.f * u);
            b[k][i] = (_Float16)(v - 0.5f);
        }
    float v0 = (float)x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
    unsigned keep = 0xffffff80u;
    asm volatile("" : "+v"(keep));
    if (threadIdx.x == 9999) pad[0] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            MF_V(c[m % 4], a[m & 3], b[(m >> 2) & 3]);
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                if (j & 1) VOP(v4, v5, v6, v7); else VOP(v0, v1, v2, v3);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int NV>
static void run_mixed(int W, int cus, float *out, unsigned long long *clk)
{
    const size_t lds = (size_t)(160 * 1024 / W) - 1024;
    auto k = mixed<NV>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 30000 / W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(cus * W), dim3(256), lds, 0, out, iters, clk);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) { best = ms; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost); }
    }
    const double mf_simd = (double)iters * 16.0 * W;
    const double ns = best * 1e6 / mf_simd;
    const double mhz = (double)h[0] / ((double)h[1] * 10.0) * 1000.0;
    printf("mixed    W=%d  %2d VALU per MFMA: %6.2f ns/MFMA/SIMD  %5.1f cyc/MFMA/SIMD  shader clock %4.0f MHz  %5.3f PFLOP/s   (VALU alone would take %d x 4 = %d cyc)\n", W, NV, ns,
           ns * mhz * 1e-3, mhz, (double)cus * 4.0 * mf_simd * 32768.0 / (best * 1e-3) / 1e15, NV, NV * 4);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

// Third question: the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64: 131 072 FLOP per instruction) by operand format -- what the
// compensated mode's correction MFMAs cost.  cbsz / blgp: 0 = fp8 e4m3, 2 = fp6 e2m3, 4 = fp4 e2m1.  Four chains, random operand bits.
// (inline asm: through the builtin hipcc moved every accumulator through AGPRs and back once per iteration)
template <int N> struct VecI { typedef int type __attribute__((ext_vector_type(N))); };
template <int F> struct FmtRegs { static constexpr int n = F == 0 ? 8 : (F == 2 ? 6 : 4); };
#define SC_STR2(x) #x
#define SC_STR(x) SC_STR2(x)
template <int FA, int FB>
__global__ __launch_bounds__(256) void scaled(float *out, int iters, unsigned long long *clk)
{
    extern __shared__ unsigned char pad[];
    typedef typename VecI<FmtRegs<FA>::n>::type va_t;
    typedef typename VecI<FmtRegs<FB>::n>::type vb_t;
    f32x16 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) c[q][r] = 0.f;
    va_t A[4];
    vb_t B[4];
    unsigned x = (threadIdx.x + 256u * blockIdx.x) * 2654435761u + 777u;
    for (int k = 0; k < 4; ++k) {
        for (int i = 0; i < FmtRegs<FA>::n; ++i) { x = x * 1664525u + 1013904223u; A[k][i] = (int)(x & 0x37373737u); }   // (exponent fields kept small: finite in every format)
        for (int i = 0; i < FmtRegs<FB>::n; ++i) { x = x * 1664525u + 1013904223u; B[k][i] = (int)(x & 0x37373737u); }
    }
    int sc = 0x7f7f7f7f;
    asm volatile("" : "+v"(sc));
    if (threadIdx.x == 9999) pad[0] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (FA == 0 && FB == 0) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(c[m % 4]) : "v"(A[m & 3]), "v"(B[(m >> 2) & 3]), "v"(sc));
            if (FA == 0 && FB == 2) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] blgp:2" : "+v"(c[m % 4]) : "v"(A[m & 3]), "v"(B[(m >> 2) & 3]), "v"(sc));
            if (FA == 2 && FB == 2) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:2 blgp:2" : "+v"(c[m % 4]) : "v"(A[m & 3]), "v"(B[(m >> 2) & 3]), "v"(sc));
            if (FA == 4 && FB == 4) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c[m % 4]) : "v"(A[m & 3]), "v"(B[(m >> 2) & 3]), "v"(sc));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += c[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}
template <int FA, int FB>
static void run_scaled(int W, int cus, float *out, unsigned long long *clk, const char *name)
{
    const size_t lds = (size_t)(160 * 1024 / W) - 1024;
    auto k = scaled<FA, FB>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 30000 / W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(cus * W), dim3(256), lds, 0, out, iters, clk);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) { best = ms; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost); }
    }
    const double mf_simd = (double)iters * 16.0 * W;
    const double ns = best * 1e6 / mf_simd;
    const double mhz = (double)h[0] / ((double)h[1] * 10.0) * 1000.0;
    printf("scaled   W=%d %-11s: %6.2f ns/MFMA/SIMD  %5.1f cyc/MFMA/SIMD  shader clock %4.0f MHz  %5.3f PFLOP/s (K = 64)\n", W, name, ns, ns * mhz * 1e-3, mhz,
           (double)cus * 4.0 * mf_simd * 131072.0 / (best * 1e-3) / 1e15);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template <int CHAINS, bool AGPR>
static void run(int W, int data, int cus, float *out, unsigned long long *clk, const char *dname)
{
    const size_t lds = (size_t)(160 * 1024 / W) - 1024;          // exactly W blocks per CU
    auto k = peak<CHAINS, AGPR>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 60000 / W;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(cus * W), dim3(256), lds, 0, out, iters, data, clk);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) { best = ms; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost); }
    }
    const double mf_simd = (double)iters * 16.0 * W;              // MFMAs one SIMD issued
    const double ns = best * 1e6 / mf_simd;
    const double mhz = (double)h[0] / ((double)h[1] * 10.0) * 1000.0;     // s_memrealtime ticks at 100 MHz
    const double pf = (double)cus * 4.0 * mf_simd * 32768.0 / (best * 1e-3) / 1e15;
    printf("%-8s W=%d chains=%d acc=%s: %7.3f ms  %6.2f ns/MFMA/SIMD  %5.1f cyc/MFMA/SIMD  shader clock %4.0f MHz  %5.3f PFLOP/s  (%4.1f %% of 2.5)\n", dname, W,
           CHAINS, AGPR ? "agpr" : "vgpr", best, ns, ns * mhz * 1e-3, mhz, pf, pf / 2.5 * 100.0);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main(int argc, char **argv)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    float *out;
    unsigned long long *clk;
    (void)hipMalloc(&out, (size_t)cus * 4 * 256 * sizeof(float));
    (void)hipMalloc(&clk, 16);
    const char *names[3] = {"zeros", "relu", "uniform"};
    const bool quick = argc > 1 && !strcmp(argv[1], "quick");
    if (argc > 1 && !strcmp(argv[1], "scaled")) {
        for (int W = 1; W <= 2; ++W) {
            run_scaled<0, 0>(W, cus, out, clk, "fp8 x fp8"); run_scaled<0, 2>(W, cus, out, clk, "fp8 x fp6"); run_scaled<2, 2>(W, cus, out, clk, "fp6 x fp6");
            run_scaled<4, 4>(W, cus, out, clk, "fp4 x fp4");
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "mixed")) {
        for (int W = 1; W <= 2; ++W) {
            run_mixed<0>(W, cus, out, clk); run_mixed<2>(W, cus, out, clk); run_mixed<4>(W, cus, out, clk); run_mixed<6>(W, cus, out, clk);
            run_mixed<8>(W, cus, out, clk); run_mixed<10>(W, cus, out, clk); run_mixed<12>(W, cus, out, clk); run_mixed<16>(W, cus, out, clk);
        }
        return 0;
    }
    for (int data = 0; data < 3; ++data) {
        for (int W = 1; W <= 4; W *= 2) {
            if (quick && W != 2) continue;
            run<2, false>(W, data, cus, out, clk, names[data]);
            run<4, false>(W, data, cus, out, clk, names[data]);
            if (W <= 2) run<8, false>(W, data, cus, out, clk, names[data]);       // (8 chains = 128 accumulator registers: two waves per SIMD at most)
            run<4, true>(W, data, cus, out, clk, names[data]);
            if (W == 1) run<8, true>(W, data, cus, out, clk, names[data]);             // (132 + 128 registers: one wave per SIMD)
        }
    }
    return 0;
}
