// Probe: what does ONE wave64 VALU instruction of the epilogues cost on gfx950, in shader cycles of its SIMD -- alone (two waves per SIMD
// issuing the same stream) and beside a partner wave on the same SIMD that issues nothing but v_mfma_f32_32x32x16_f16?
// (Round 5: the fused stem's phase 1 takes ~11 k cycles per tile where an issue model of 4 cycles per VALU instruction and 32 per MFMA
//  predicts 5-7.5 k; the candidates are the packed fp32 operations, the SDWA conversions and the 32-value fp6 conversion.)
// Every kind is a straight-line block of 16 INDEPENDENT instances (no dependent-issue stalls), repeated `iters` times.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int v6i __attribute__((ext_vector_type(6)));

enum { K_FMA, K_PK_FMA, K_PK_ADD, K_PK_MUL, K_MED3, K_MAX3, K_CVT_PK_F16, K_CVT_F32_F16, K_CVT_F32_F16_SDWA, K_FMA_MIX, K_FMA_MIX_HI, K_CVT_FP6,
       K_PK_ADD_F16, K_PK_MAX_F16, K_CVT_PKRTZ, K_PERM, K_AND, K_MOV, K_MUL, K_ADD, K_MAX, K_CNDMASK, K_LSHL_ADD, K_PK_MOV, K_MAX_F16, K_FMA_MIXLO, K_CVT_SC_FP8_F16, K_CVT_PK_FP8_F32, K_CVT_F32_FP8, K_CVT_F32_FP8_SDWA, K_CVT_F16_F32, K_PERMLANE32_SWAP, K_AND_OR, K_XOR, K_ADD_U32, K_CNDMASK_E64, K_CMP_CND, K_MAX_I32_DPP, K_CVT_SC_PK_F16_FP8, K_PK_MUL_F16, K_DOT2, K_MIN3, K_CVT_SC_PK32_FP6_F16, K_BFE, K_N };
static const char *kind_name[K_N] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_med3_f32", "v_max3_f32", "v_cvt_pk_f16_f32",
    "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa(WORD_1)", "v_fma_mix_f32 (f16 lo,f32,f32)", "v_fma_mix_f32 (f16 hi,f32,f32)", "v_cvt_scalef32_2xpk16_fp6_f32",
    "v_pk_add_f16", "v_pk_max_f16", "v_cvt_pkrtz_f16_f32", "v_perm_b32", "v_and_b32", "v_mov_b32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32",
    "v_lshl_add_u32", "v_pk_mov_b32", "v_max_f16", "v_fma_mixlo_f16",
    "v_cvt_scalef32_pk_fp8_f16", "v_cvt_pk_fp8_f32", "v_cvt_f32_fp8", "v_cvt_f32_fp8_sdwa(BYTE_2)", "v_cvt_f16_f32", "v_permlane32_swap_b32", "v_and_or_b32", "v_xor_b32", "v_add_u32", "v_cndmask_b32_e64 (SGPR pair)",
    "v_cmp_gt_f32 + v_cndmask (vcc)", "v_max_i32_dpp row_shr:1", "v_cvt_scalef32_pk_f16_fp8", "v_pk_mul_f16", "v_dot2c_f32_f16", "v_min3_f32", "v_cvt_scalef32_pk32_fp6_f16", "v_bfe_u32"};

template <int KIND>
__device__ __forceinline__ void body(float (&x)[16], float (&y)[16], f32x2 (&p)[16], f32x2 (&q)[16], f32x16 &big0, f32x16 &big1, v6i (&d6)[4], float s)
{
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(q[(i + 1) & 15]));
        if (KIND == K_PK_ADD) asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[i]) : "v"(q[i]));
        if (KIND == K_PK_MUL) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(q[i]));
        if (KIND == K_MED3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(x[i]) : "v"(s));
        if (KIND == K_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_CVT_PK_F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_CVT_F32_F16) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_F32_F16_SDWA) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_FMA_MIX) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_FMA_MIX_HI) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_CVT_FP6 && i < 4) asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d6[i]) : "v"(big0), "v"(big1), "v"(s));
        if (KIND == K_PK_ADD_F16) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_PKRTZ) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
        if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : );
        if (KIND == K_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_PK_MOV) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "=v"(p[i]) : "v"(q[i]));
        if (KIND == K_MAX_F16) asm volatile("v_max_f16 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_SC_FP8_F16) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_CVT_PK_FP8_F32) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_CVT_F32_FP8) asm volatile("v_cvt_f32_fp8_e32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_F32_FP8_SDWA) asm volatile("v_cvt_f32_fp8_sdwa %0, %1 src0_sel:BYTE_2" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_F16_F32) asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
        if (KIND == K_PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(x[i]), "+v"(y[i]));
        if (KIND == K_AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(x[i]) : "v"(y[i]) : "s20", "s21");
        if (KIND == K_CMP_CND) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
        if (KIND == K_MAX_I32_DPP) asm volatile("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_CVT_SC_PK_F16_FP8) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(x[i]) : "v"(y[i]), "v"(s));
        if (KIND == K_PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
        if (KIND == K_DOT2) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_MIN3) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 15]));
        if (KIND == K_CVT_SC_PK32_FP6_F16 && i < 4) asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(d6[i]) : "v"(big0), "v"(s));
        if (KIND == K_BFE) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x[i]));
        if (KIND == K_FMA_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(x[i]) : "v"(y[i]), "v"(s), "v"(y[(i + 1) & 15]));
    }
}

// block = 512 threads = two waves per SIMD.  partner == 0: all eight waves run the VALU stream.  partner == 1: waves 4..7 issue only MFMAs
// (their count is reported so that the MFMA rate beside the stream can be read too).
template <int KIND>
__global__ __launch_bounds__(512) void probe(float *out, int iters, int partner, unsigned long long *clk)
{
    extern __shared__ unsigned char pad[];
    const int wave = threadIdx.x >> 6;
    float x[16], y[16];
    f32x2 p[16], q[16];
    f32x16 big0, big1;
    v6i d6[4];
    unsigned r = (threadIdx.x + 512u * blockIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 16; ++i) {
        r = r * 1664525u + 1013904223u;
        x[i] = (float)(r >> 16) / 65536.f; y[i] = (float)(r & 0xffff) / 65536.f + 0.25f;
        p[i] = f32x2{x[i], y[i]}; q[i] = f32x2{y[i], x[i]};
        big0[i] = x[i]; big1[i] = y[i];
    }
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 6; ++k) d6[i][k] = 0;
    if (threadIdx.x == 9999) pad[0] = 1;
    float s = 0.999f;
    asm volatile("" : "+v"(s));
    f32x16 c[4];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) c[k][e] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(x[i] - 0.5f); b[i] = (_Float16)((i & 1) ? 0.f : y[i]); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (partner && wave >= 4) {
        // MFMA-only partner: runs for about as long as the VALU waves (16 instances x 4 cycles ~ 2 MFMAs per body)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 2; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
        }
    } else {
        for (int it = 0; it < iters; ++it) body<KIND>(x, y, p, q, big0, big1, d6, s);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.f;
    for (int i = 0; i < 16; ++i) acc += x[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 6; ++k) acc += (float)d6[i][k];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc += c[k][e];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
}

template <int KIND>
static void run(float *d, unsigned long long *clk, int cus)
{
    const int iters = 4000;
    const int n_inst = (KIND == K_CVT_FP6 || KIND == K_CVT_SC_PK32_FP6_F16) ? 4 : 16;
    for (int partner = 0; partner < 2; ++partner) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        probe<KIND><<<cus, 512, 150 * 1024>>>(d, iters, partner, clk);   // warm
        probe<KIND><<<cus, 512, 150 * 1024>>>(d, iters, partner, clk);
        hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double valu = 0, mf = 0;
        const int nv = partner ? 4 : 8;
        for (int w = 0; w < nv; ++w) valu += (double)h[w] / nv;
        for (int w = 4; w < 8; ++w) mf += (double)h[w] / 4;
        // waves per SIMD issuing the stream: 2 (partner 0) or 1 (partner 1)
        const double per_inst_wave = valu / ((double)iters * n_inst);
        if (!partner)
            printf("%-34s two VALU waves/SIMD: %6.2f cycles per instruction and SIMD (%.2f per wave's own instruction)\n", kind_name[KIND], per_inst_wave / 2, per_inst_wave);
        else
            printf("%-34s beside an MFMA-only wave: %6.2f cycles per instruction; partner: %6.1f cycles per MFMA\n", kind_name[KIND], per_inst_wave, mf / ((double)iters * 2));
    }
}

template <int K>
static void run_all(float *d, unsigned long long *clk, int cus)
{
    if constexpr (K < K_N) {
        run<K>(d, clk, cus);
        run_all<K + 1>(d, clk, cus);
    }
}

int main()
{
    float *d;
    unsigned long long *clk;
    hipMalloc(&d, 512 * 512 * 4);
    hipMalloc(&clk, 64);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    run_all<0>(d, clk, cus);
    return 0;
}
