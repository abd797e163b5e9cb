// Internal declarations shared by the HIP translation units of libsfd2hip.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

// Experiment / ablation switches are read from the environment only in builds made with -DSFD2_EXPERIMENTS
// (tools/ A/B runs); the product .so ignores them, so an exported variable can never change its results.
#ifdef SFD2_EXPERIMENTS
#include <stdlib.h>
static inline const char *sfd2_env(const char *name) { return getenv(name); }
#else
static inline const char *sfd2_env(const char *) { return nullptr; }
#endif

// Option "cu_limit" (per context): the persistent kernels launch at most this many blocks (0 = one per CU).  With two streams per
// GPU and half the CUs each, two DIFFERENT kernels (of two images) run side by side instead of taking turns on the whole chip.
// The launchers read a thread-local that every network pass sets from ITS context (run_network, api_network.hip): launches are
// made on the caller's thread, so two contexts never see each other's limit.
extern thread_local int g_sfd2_cu_limit;
static inline int sfd2_slots(int cus) { return (g_sfd2_cu_limit > 0 && g_sfd2_cu_limit < cus) ? g_sfd2_cu_limit : cus; }

// Block barrier that must make other waves' LDS-DMA copies (global_load_lds) visible: the drain of the vector-memory
// counter is written out.  A plain __syncthreads() happens to emit the same s_waitcnt vmcnt(0) today while a copy is in
// flight, but that is hipcc's choice, not a language guarantee.
#define SFD2_BARRIER_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

typedef _Float16 half_t;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#ifdef __HIPCC__
// 16-byte LDS read of four floats, TYPED LIKE THE MFMA FRAGMENT READS (_Float16 x 8).  hipcc orders LDS reads against
// in-flight direct-to-LDS copies (global_load_lds) by type-based alias analysis: a float-typed ds_read "may alias" the
// copies' destination and gets an s_waitcnt vmcnt(0) in front of it, which drains every copy (and store) in flight;
// read as the fragments' type, the same bytes carry no such wait.  Use for scale / shift tables that share LDS with
// staged tiles in kernels that keep copies in flight across the read (conv1x1_c256_kernel, resblock_kernel).
__device__ __forceinline__ float4 sfd2_lds_f4(const float *p)
{
    const h8_t raw = *reinterpret_cast<const h8_t *>(reinterpret_cast<const unsigned char *>(p));
    float4 r;
    __builtin_memcpy(&r, &raw, 16);
    return r;
}
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ compensated fp16 ("f16c", SFD2_PREC_F16C)
// A compensated tensor is TWO planes of 2-byte units with the same NHWC geometry: `hi` = fp16(x) and `corr`, whose unit
// for a channel is the byte pair (fp8 e4m3 of (x - hi) * 2^9, fp8 e4m3 of x * 2^-2).  A compensated filter array is the
// fp16 filters followed by the same number of units (fp8 of w * 2^b0, fp8 of (w - fp16(w)) * 2^(b0 + 11)), b0 chosen per
// layer so that max|w| * 2^b0 <= 448.  A compensated layer then computes, for every tap and 32-channel chunk,
//     acc += hi_x . hi_w                      two v_mfma_f32_32x32x16_f16          (K = 32 channels)
//     acc += corr_x . corr_w * 2^-(9 + b0)     ONE v_mfma_scale_f32_32x32x64_f8f6f4 (K = 32 channels x 2 terms:
//                                              lo_x * w  +  x * lo_w, byte-wise pairing of the two units)
// i.e. the two first-order rounding terms of the fp16 product at fp8 precision: both operands then carry ~15
// significant bits instead of 11 for 1.65x the matrix time of the plain fp16 layer (a hi / lo fp16 split needs 3x).
// The scale is uniform over K (E8M0 byte 127 - 9 - b0 on the A operand), so no block layout of the scales matters; and
// because A and B map lane bytes to K identically, any arrangement of a chunk's 64 bytes is valid as long as filters
// and pixels use the same one -- the corr records are read with exactly the addresses of the fp16 records.
// Compensated tensors SATURATE at +-1792 (SFD2_C_SAT; the plain fp16 path: +-65504): with the value bounded neither corr
// byte can overflow fp8's 448, and the bound costs nothing (it is the third operand of the ReLU's v_med3_f32).
#define SFD2_C_XL_SHIFT 9            // corr unit byte 0: fp8((x - hi) * 2^9)
#define SFD2_C_XH_SCALE 0.25f        // corr unit byte 1: fp8(x * 2^-2)
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
#ifdef __HIPCC__
__device__ __forceinline__ v8i_t sfd2_cat8(h8_t a, h8_t b)
{
    v4i_t x, y;
    __builtin_memcpy(&x, &a, 16);
    __builtin_memcpy(&y, &b, 16);
    return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
}
// K slice kk (0 / 1) of an 8-dword fragment tuple, as the fp16 MFMA's operand (a sub-register reference, no copy)
__device__ __forceinline__ h8_t sfd2_half8(v8i_t v, int kk)
{
    const v4i_t x = kk ? __builtin_shufflevector(v, v, 4, 5, 6, 7) : __builtin_shufflevector(v, v, 0, 1, 2, 3);
    h8_t r;
    __builtin_memcpy(&r, &x, 16);
    return r;
}
// fp8 MFMA of one 32-channel corr chunk: a0 / a1 and b0 / b1 are the two 16-byte fragments the fp16 path would feed to
// its two K = 16 MFMAs; sa = the layer's scale byte replicated into all four bytes
__device__ __forceinline__ f32x16_t sfd2_mfma_corr(h8_t a0, h8_t a1, h8_t b0, h8_t b1, f32x16_t acc, int sa)
{
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(sfd2_cat8(a0, a1), sfd2_cat8(b0, b1), acc, 0, 0, 0, sa, 0, 0x7f7f7f7f);
}
// The same instruction with the accumulator TIED (destination = C operand), by inline asm.  To instruction selection the scaled MFMA's
// destination is a fresh value: in straight-line code the accumulators then ping-pong between two register sets for free, but where two
// control-flow paths (an fp16 unit / an fp8 unit behind a wave-uniform flag) meet, hipcc copies every accumulator back to its home
// registers -- conv3x3_rf<2, 128, comp> carried 160 v_mov_b64 per chunk, as much VALU time as its MFMAs (profiles/r04_conv2b_ablations.txt).
// The asm is opaque to hipcc's hazard recogniser: a VALU read of the accumulators must be preceded by sfd2_mfma_settle().
__device__ __forceinline__ void sfd2_mfma_corr_tied(f32x16_t &acc, v8i_t a, v8i_t b, int sa, int sb)
{
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
// One unit (tap x 32 channels) of a kernel whose chunk type is a wave-uniform RUNTIME flag, for two pixel fragments: four fp16 MFMAs (both K
// slices of both fragments) or two scaled fp8 MFMAs, chosen by a scalar branch INSIDE the asm, so that to hipcc both chunk types are the same
// straight-line statement with tied accumulators.  b0 / b1: the fragments' 8-dword tuples (their halves are the fp16 operands), a8 = (a0, a1).
__device__ __forceinline__ void sfd2_mfma_unit2(f32x16_t &c0, f32x16_t &c1, h8_t a0, h8_t a1, v8i_t a8, v8i_t b0, v8i_t b1, int sa, int sb, int f8)
{
    const h8_t b00 = sfd2_half8(b0, 0), b01 = sfd2_half8(b0, 1), b10 = sfd2_half8(b1, 0), b11 = sfd2_half8(b1, 1);
    asm volatile("s_cmp_lg_u32 %11, 0\n\t"
                 "s_cbranch_scc1 .Lsfd2_u8_%=\n\t"
                 "v_mfma_f32_32x32x16_f16 %0, %2, %5, %0\n\t"
                 "v_mfma_f32_32x32x16_f16 %1, %2, %7, %1\n\t"
                 "v_mfma_f32_32x32x16_f16 %0, %3, %6, %0\n\t"
                 "v_mfma_f32_32x32x16_f16 %1, %3, %8, %1\n\t"
                 "s_branch .Lsfd2_ue_%=\n"
                 ".Lsfd2_u8_%=:\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %0, %4, %12, %0, %9, %10 op_sel_hi:[0,0,0]\n\t"
                 "v_mfma_scale_f32_32x32x64_f8f6f4 %1, %4, %13, %1, %9, %10 op_sel_hi:[0,0,0]\n"
                 ".Lsfd2_ue_%=:"
                 : "+v"(c0), "+v"(c1)
                 : "v"(a0), "v"(a1), "v"(a8), "v"(b00), "v"(b01), "v"(b10), "v"(b11), "v"(sa), "v"(sb), "s"(f8), "v"(b0), "v"(b1)
                 : "scc");
}
__device__ __forceinline__ void sfd2_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }   // 32 wait states >= any XDL write -> VALU read distance
// corr units of two channels (values v0, v1 with fp16 parts h0, h1) as one dword: bytes (lo8_0, x8_0, lo8_1, x8_1).
// v_cvt_pk_fp8_f32 returns NaN above 464, hence the clamps (x saturates at 1792, its residual then too).
__device__ __forceinline__ unsigned sfd2_corr2(float v0, half_t h0, float v1, half_t h1)
{
    const float l0 = __builtin_amdgcn_fmed3f((v0 - (float)h0) * (float)(1 << SFD2_C_XL_SHIFT), -448.0f, 448.0f);
    const float l1 = __builtin_amdgcn_fmed3f((v1 - (float)h1) * (float)(1 << SFD2_C_XL_SHIFT), -448.0f, 448.0f);
    const float x0 = __builtin_amdgcn_fmed3f(v0 * SFD2_C_XH_SCALE, -448.0f, 448.0f);
    const float x1 = __builtin_amdgcn_fmed3f(v1 * SFD2_C_XH_SCALE, -448.0f, 448.0f);
    int d = __builtin_amdgcn_cvt_pk_fp8_f32(l0, x0, 0, false);
    d = __builtin_amdgcn_cvt_pk_fp8_f32(l1, x1, d, true);
    return (unsigned)d;
}
// hi + corr planes of four consecutive channels: hv (8 bytes of fp16) and cv (8 bytes of corr units)
__device__ __forceinline__ void sfd2_split4(float v0, float v1, float v2, float v3, uint2 &hv, uint2 &cv)
{
    h4_t h;
    h[0] = (half_t)v0; h[1] = (half_t)v1; h[2] = (half_t)v2; h[3] = (half_t)v3;
    __builtin_memcpy(&hv, &h, 8);
    cv.x = sfd2_corr2(v0, h[0], v1, h[1]);
    cv.y = sfd2_corr2(v2, h[2], v3, h[3]);
}
// The epilogue of a compensated layer for four consecutive channels, written for the VALU budget (the compensated stem's
// conv1a phase was bound by it: 216 VALU per 16 outputs): y = acc * scale + shift (+ add) as packed-fp32 FMAs / adds
// (v_pk_fma_f32 / v_pk_add_f32), ONE v_med3_f32 per value that is both the ReLU (lo = 0; lo = -SFD2_C_SAT without) and the
// saturation of the compensated tensors at +-1792 -- with the value bounded, its fp16 residual (<= 0.5) and its quarter
// (<= 448) need no clamps of their own in front of v_cvt_pk_fp8_f32 (which returns NaN above 464) -- then packed multiplies
// for the two unit scalings.  Identical results to sfd2_split4 behind a scalar epilogue for |y| <= 1792.
#define SFD2_C_SAT 1792.0f
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sfd2_corr2v(f32x2_t v, h2_t h)
{
    const f32x2_t hf = {(float)h[0], (float)h[1]};
    const f32x2_t l = (v - hf) * (float)(1 << SFD2_C_XL_SHIFT);
    const f32x2_t x = v * SFD2_C_XH_SCALE;
    int d = __builtin_amdgcn_cvt_pk_fp8_f32(l[0], x[0], 0, false);
    d = __builtin_amdgcn_cvt_pk_fp8_f32(l[1], x[1], d, true);
    return (unsigned)d;
}
// ---- range status: the largest value every compensated layer WOULD have stored, before the saturation.
// A lane folds its values into `mx` in the epilogue (one v_max3_f32 per two values); sfd2_wave_max_bits reduces the wave with
// six DPP steps and hands back a wave-uniform word (post-ReLU values are >= 0, so the bit patterns order like the floats) that a
// kernel keeps in a SCALAR register across its tiles -- no vector register lives through the K loops for it -- and every wave
// commits once, when it exits, into one of SFD2_RANGE_SUB sub-slots of the tensor's slot (blocks spread over them so that
// 2 048 waves do not queue on one address).  sfd2_get_range_status folds the sub-slots on the host.
#define SFD2_RANGE_SUB 16
#define SFD2_ZERO_PAGE_BYTES 1024       // the context's zero page; the range-status words follow it in the same allocation
enum { SFD2_RS_CONV1A, SFD2_RS_CONV1B, SFD2_RS_CONV2A, SFD2_RS_CONV2B, SFD2_RS_CONV3A, SFD2_RS_CONV3B, SFD2_RS_T1_0, SFD2_RS_T1_1, SFD2_RS_T1_2,
       SFD2_RS_T2_0, SFD2_RS_T2_1, SFD2_RS_T2_2, SFD2_RS_OUT_0, SFD2_RS_OUT_1, SFD2_RS_OUT_2, SFD2_RS_PA0, SFD2_RS_DA0, SFD2_RS_COUNT };
__device__ __forceinline__ float sfd2_max3(float m, float a, float b) { return __builtin_fmaxf(__builtin_fmaxf(m, a), b); }
__device__ __forceinline__ unsigned int sfd2_wave_max_bits(float mx)
{
#ifdef SFD2_NO_RANGE      // timing experiment (tools/ab_libs.py): what the recording costs
    return 0u;
#endif
    int v = __float_as_int(__builtin_fmaxf(mx, 0.0f));
#define SFD2_DPP_MAX(ctrl_, rmask_) { const int o_ = __builtin_amdgcn_update_dpp(0, v, ctrl_, rmask_, 0xf, false); v = o_ > v ? o_ : v; }
    SFD2_DPP_MAX(0x111, 0xf)      // row_shr:1   (lanes without a source keep `old` = 0: the identity here)
    SFD2_DPP_MAX(0x112, 0xf)      // row_shr:2
    SFD2_DPP_MAX(0x114, 0xf)      // row_shr:4
    SFD2_DPP_MAX(0x118, 0xf)      // row_shr:8   -> lane 15 of every row holds the row's maximum
    SFD2_DPP_MAX(0x142, 0xa)      // row_bcast:15 into rows 1 and 3
    SFD2_DPP_MAX(0x143, 0xc)      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
#undef SFD2_DPP_MAX
    return (unsigned int)__builtin_amdgcn_readlane(v, 63);
}
// One commit per wave, at its exit.  The word is READ first and the atomic issued only when this wave has something larger to say:
// the words are sticky (running maxima until the host resets them), so from the second image on nearly no wave writes.  That matters:
// device-scope atomics of 2 048 waves on a handful of addresses are resolved one after the other at the memory side -- measured
// +10 .. +17 us at the tail of every recording kernel when each wave simply issued its atomic (profiles/r04_range_cost.txt).
// (The read may come from this XCD's L2 and be stale, i.e. too small: then the atomic is issued needlessly, never wrongly skipped
// for a value that is not already recorded -- a larger recorded value only ever makes the skip right.)
__device__ __forceinline__ void sfd2_range_commit(unsigned int *slot /* the tensor's SFD2_RANGE_SUB words, or null */, unsigned int wave_bits)
{
    if (slot == nullptr || wave_bits == 0u) return;      // (wave-uniform)
    unsigned int *w = slot + (blockIdx.x & (SFD2_RANGE_SUB - 1));
    if ((threadIdx.x & 63) == 0) {
        const unsigned int cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave_bits > cur) atomicMax(w, wave_bits);
    }
}

// the same record taken from the STORED fp16 values (after the saturation: it reads 1792 when values were clamped, which is all
// the status needs): for kernels with no vector register to spare in their epilogue -- the operands are the packed words that
// are about to be stored, one v_pk_max_f16 per two values
__device__ __forceinline__ void sfd2_track_h4(uint2 hv, unsigned int &mxh)
{
    h2_t a, b, m;
    __builtin_memcpy(&a, &hv.x, 4);
    __builtin_memcpy(&b, &hv.y, 4);
    __builtin_memcpy(&m, &mxh, 4);
    m = __builtin_elementwise_max(__builtin_elementwise_max(m, a), b);
    __builtin_memcpy(&mxh, &m, 4);
}
__device__ __forceinline__ float sfd2_h2_max(unsigned int mxh)
{
    h2_t m;
    __builtin_memcpy(&m, &mxh, 4);
    return __builtin_fmaxf((float)m[0], (float)m[1]);
}

template <bool ADD, bool TRACK = true>
__device__ __forceinline__ void sfd2_epi4(float a0, float a1, float a2, float a3, float4 sc, float4 sh, float4 add, float lo,
                                          uint2 &hv, uint2 &cv, float &mx, bool counted = true /* false: a lane past the image's edge */)
{
    f32x2_t v01 = f32x2_t{a0, a1} * f32x2_t{sc.x, sc.y} + f32x2_t{sh.x, sh.y};
    f32x2_t v23 = f32x2_t{a2, a3} * f32x2_t{sc.z, sc.w} + f32x2_t{sh.z, sh.w};
    if (ADD) { v01 += f32x2_t{add.x, add.y}; v23 += f32x2_t{add.z, add.w}; }
#ifndef SFD2_NO_RANGE
    if (TRACK) {
        const float m = sfd2_max3(sfd2_max3(mx, v01[0], v01[1]), v23[0], v23[1]);
        mx = counted ? m : mx;
    }
#endif
    v01[0] = __builtin_amdgcn_fmed3f(v01[0], lo, SFD2_C_SAT); v01[1] = __builtin_amdgcn_fmed3f(v01[1], lo, SFD2_C_SAT);
    v23[0] = __builtin_amdgcn_fmed3f(v23[0], lo, SFD2_C_SAT); v23[1] = __builtin_amdgcn_fmed3f(v23[1], lo, SFD2_C_SAT);
    const h2_t h01 = __builtin_convertvector(v01, h2_t), h23 = __builtin_convertvector(v23, h2_t);      // (one v_cvt_pk_f16_f32 per pair, round to nearest even)
    __builtin_memcpy(&hv.x, &h01, 4);
    __builtin_memcpy(&hv.y, &h23, 4);
    cv.x = sfd2_corr2v(v01, h01);
    cv.y = sfd2_corr2v(v23, h23);
}
// ... the same with ONE correction byte per channel (the residual; option "trunk_r1": rb23_c_kernel.hip): rv = the four residual bytes
template <bool ADD, bool TRACK = true>
__device__ __forceinline__ void sfd2_epi4_r1(float a0, float a1, float a2, float a3, float4 sc, float4 sh, float4 add, float lo,
                                             uint2 &hv, unsigned int &rv, float &mx, bool counted = true)
{
    f32x2_t v01 = f32x2_t{a0, a1} * f32x2_t{sc.x, sc.y} + f32x2_t{sh.x, sh.y};
    f32x2_t v23 = f32x2_t{a2, a3} * f32x2_t{sc.z, sc.w} + f32x2_t{sh.z, sh.w};
    if (ADD) { v01 += f32x2_t{add.x, add.y}; v23 += f32x2_t{add.z, add.w}; }
#ifndef SFD2_NO_RANGE
    if (TRACK) {
        const float m = sfd2_max3(sfd2_max3(mx, v01[0], v01[1]), v23[0], v23[1]);
        mx = counted ? m : mx;
    }
#endif
    v01[0] = __builtin_amdgcn_fmed3f(v01[0], lo, SFD2_C_SAT); v01[1] = __builtin_amdgcn_fmed3f(v01[1], lo, SFD2_C_SAT);
    v23[0] = __builtin_amdgcn_fmed3f(v23[0], lo, SFD2_C_SAT); v23[1] = __builtin_amdgcn_fmed3f(v23[1], lo, SFD2_C_SAT);
    const h2_t h01 = {(half_t)v01[0], (half_t)v01[1]}, h23 = {(half_t)v23[0], (half_t)v23[1]};      // (element-wise: as a vector conversion conv3x3_pp<.., 307> spills)
    __builtin_memcpy(&hv.x, &h01, 4);
    __builtin_memcpy(&hv.y, &h23, 4);
    const f32x2_t l01 = (v01 - f32x2_t{(float)h01[0], (float)h01[1]}) * (float)(1 << SFD2_C_XL_SHIFT);
    const f32x2_t l23 = (v23 - f32x2_t{(float)h23[0], (float)h23[1]}) * (float)(1 << SFD2_C_XL_SHIFT);
    int d = __builtin_amdgcn_cvt_pk_fp8_f32(l01[0], l01[1], 0, false);
    d = __builtin_amdgcn_cvt_pk_fp8_f32(l23[0], l23[1], d, true);
    rv = (unsigned int)d;
}
// ---- corr records in fp6 (round 4; option "fp6_acts": the tensors whose only readers are conv3x3_pp layers).  The scaled MFMA takes 66
// cycles with fp8 operands on either side and 33.5 with e2m3 on both (profiles/r04_mfma_probe.txt), so the correction operands of those
// layers are block-scaled e2m3: per pixel and lane half a HALF-RECORD of 32 bytes = 32 six-bit codes (24 B) + one E8M0 scale byte (replicated
// into a dword) + 4 B padding, occupying the 16-byte slots {lhi, 2 + lhi} of the pixel's 64-byte record -- exactly the two slots the
// consumer's lane half reads today, so staging and fragment reads do not change; the fragment's dwords 0..5 are the operand, dword 6 the
// scale.  Codes (v_cvt_scalef32_2xpk16_fp6_f32, probed in tools/probe/cvt_fp6.hip: code 2 j = src0[j], code 2 j + 1 = src1[j], round to
// nearest even, saturating at 7.5, value / scale): 2 j = lo'_j = (x_j - fp16(x_j)) * 2^11, 2 j + 1 = x_j, for the lane's 16 channels j = 4 q + r
// <-> channel 8 q + 4 lhi + r of the chunk (the MFMA C layout of the producer).  |lo'| <= x, so ONE scale 2^E >= max x / 7.5 serves both;
// the filters' strings carry (w, lo'_w) in the same positions and 2^-11 in their scale byte.  No fixed shifts: every block has its exponent.
// SFD2_PIX6_BF6 = 1: the PIXEL codes are bf6 (e3m2: largest value 28, subnormal step 2^-4) instead of fp6 (e2m3: 7.5, 2^-3): one mantissa
// bit less on the block's largest values, three binades more under the block's maximum before a residual falls into the subnormal step.
#ifndef SFD2_PIX6_BF6
#define SFD2_PIX6_BF6 0
#endif
#define SFD2_PIX6_BLGP (SFD2_PIX6_BF6 ? 3 : 2)
typedef float f32x16v_t __attribute__((ext_vector_type(16)));
typedef int v6i_t __attribute__((ext_vector_type(6)));
template <bool TRACK = true>
__device__ __forceinline__ void sfd2_epi16_fp6(const f32x16_t &acc, const float4 (&sc)[4], const float4 (&sh)[4], float lo, uint2 (&hv)[4], uint4 &rec0, uint4 &rec1,
                                               float &mx, bool counted = true)
{
    f32x16v_t v, l;
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x2_t v01 = f32x2_t{acc[4 * q], acc[4 * q + 1]} * f32x2_t{sc[q].x, sc[q].y} + f32x2_t{sh[q].x, sh[q].y};
        f32x2_t v23 = f32x2_t{acc[4 * q + 2], acc[4 * q + 3]} * f32x2_t{sc[q].z, sc[q].w} + f32x2_t{sh[q].z, sh[q].w};
        m = sfd2_max3(sfd2_max3(m, v01[0], v01[1]), v23[0], v23[1]);
        v01[0] = __builtin_amdgcn_fmed3f(v01[0], lo, SFD2_C_SAT); v01[1] = __builtin_amdgcn_fmed3f(v01[1], lo, SFD2_C_SAT);
        v23[0] = __builtin_amdgcn_fmed3f(v23[0], lo, SFD2_C_SAT); v23[1] = __builtin_amdgcn_fmed3f(v23[1], lo, SFD2_C_SAT);
        const h2_t h01 = __builtin_convertvector(v01, h2_t), h23 = __builtin_convertvector(v23, h2_t);      // (one v_cvt_pk_f16_f32 per pair, round to nearest even)
        __builtin_memcpy(&hv[q].x, &h01, 4);
        __builtin_memcpy(&hv[q].y, &h23, 4);
        const f32x2_t l01 = (v01 - f32x2_t{(float)h01[0], (float)h01[1]}) * 2048.0f, l23 = (v23 - f32x2_t{(float)h23[0], (float)h23[1]}) * 2048.0f;
        v[4 * q] = v01[0]; v[4 * q + 1] = v01[1]; v[4 * q + 2] = v23[0]; v[4 * q + 3] = v23[1];
        l[4 * q] = l01[0]; l[4 * q + 1] = l01[1]; l[4 * q + 2] = l23[0]; l[4 * q + 3] = l23[1];
    }
#ifndef SFD2_NO_RANGE
    if (TRACK) mx = counted ? __builtin_fmaxf(mx, m) : mx;
#endif
    const unsigned int mb = __float_as_uint(__builtin_fminf(__builtin_fmaxf(m, 0.0f), SFD2_C_SAT));
    v6i_t d;
#if SFD2_PIX6_BF6
    // the block's exponent: the smallest E with max / 2^E <= 28 (max = f * 2^k, f in [0.5, 1): E = k - 5, one more when f > 0.875)
    int e8 = (int)(mb >> 23) - 4 + ((mb & 0x7fffffu) > 0x600000u ? 1 : 0);
    e8 = e8 < 1 ? 1 : e8;
    const float scale = __uint_as_float((unsigned int)e8 << 23);
    asm volatile("v_cvt_scalef32_2xpk16_bf6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(l), "v"(v), "v"(scale));
#else
    // the block's exponent: the smallest E with max / 2^E <= 7.5 (max = f * 2^k, f in [0.5, 1): E = k - 3, one more when f > 0.9375)
    int e8 = (int)(mb >> 23) - 2 + ((mb & 0x7fffffu) > 0x700000u ? 1 : 0);
    e8 = e8 < 1 ? 1 : e8;
    const float scale = __uint_as_float((unsigned int)e8 << 23);
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(l), "v"(v), "v"(scale));
#endif
    rec0 = make_uint4((unsigned)d[0], (unsigned)d[1], (unsigned)d[2], (unsigned)d[3]);
    rec1 = make_uint4((unsigned)d[4], (unsigned)d[5], (unsigned)e8 * 0x01010101u, 0u);
}
// the residual (x - hi) two corr units of a dword carry, as floats
__device__ __forceinline__ float sfd2_corr_lo(unsigned d, int ch /*0 or 1*/)
{
    return (ch ? __builtin_amdgcn_cvt_f32_fp8((int)d, 2) : __builtin_amdgcn_cvt_f32_fp8((int)d, 0)) * (1.0f / (float)(1 << SFD2_C_XL_SHIFT));
}
#endif

// ------------------------------------------------------------------ conv stack
// Activations live in HBM as NHWC fp16 (channel pitch = padded Cout of the producer).

// conv1a: fp32 CHW image (+ optional norm_RGB) -> 3x3 conv 3->64 + folded BN + ReLU -> NHWC fp16
void launch_conv1a(hipStream_t st, const float *img_chw, int H, int W, int normalise,
                   const half_t *wpk /*[2][3][64][8]*/, const float *scale, const float *shift,
                   half_t *out /*[H][W][64]*/);

// Fused stem: norm_RGB + conv1a + BN + ReLU + conv1b (stride 2) + BN + ReLU; w2 = [9][64 oc][64 ic] fp16
void launch_fused_stem(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *w1, const float *sc1,
                       const float *sh1, const half_t *w2, const float *sc2, const float *sh2, half_t *out, int H2, int W2);

// Implicit-GEMM conv on MFMA (3x3 or 1x1, stride 1 or 2, Cin % 32 == 0, Cout_pad % 64 == 0).
//   in  [H][W][Cin] fp16,  wpk [Cin/cc][ks*ks][Cout_pad][cc] fp16 (cc = conv_igemm_chunk),  scale/shift [Cout_pad]
//   out [Ho][Wo][Cout_pad] fp16 (relu / residual optional) or fp32 (out_f32)
// conv3_kernels.hip: the 3x3 stride-1 layers with >= 256 output channels
bool conv3x3_pp_serves(int ks, int stride, int CoutP, int Cin);
// conv3rf_kernels.hip: the same layers (and their stride-2 siblings) when the output is small
bool launch_conv3x3_rf_x3(hipStream_t st, const half_t *in_hi, const half_t *in_lo, int H, int W, int Cin, const half_t *wpl,
                          const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out_hi, half_t *out_lo,
                          float *out_f32, int Ho, int Wo, const half_t *zero_page);
bool conv3x3_rf_serves(int ks, int stride, int CoutP, int Cin, int Ho, int Wo);
void launch_conv_igemm(hipStream_t st, const half_t *in, int H, int W, int Cin,
                       const half_t *wpk, const float *scale, const float *shift, int Cout_pad,
                       int ks, int stride, int relu, const half_t *residual,
                       void *out, int out_f32, int Ho, int Wo, const half_t *zero_page /*>= 64 B of zeros*/);

// input channels per packed filter tile for this layer: wpk is [Cin/cc][ks*ks][Cout_pad][cc]
int conv_igemm_chunk(int ks, int stride, int Cout_pad, int Cin);

// Grouped 3x3 conv, 256 channels, 32 groups of 8 (ResBlock.conv2) + folded BN + ReLU.
//   wpk [16 pairs][5 steps][64 lanes][8] fp16 (block-diagonal 16x16 MFMA A fragments)
void launch_gconv3x3_g8(hipStream_t st, const half_t *in, int H, int W, const half_t *wpk,
                        const float *scale, const float *shift, half_t *out);

// 1x1 conv 256 -> 3 (ConvSta), fp32 planar output [3][H][W]
void launch_convsta(hipStream_t st, const half_t *in, int npix, const float *w /*[3][256]*/, const float *b,
                    float *out /*[3][npix]*/);

// ---- compensated fp16 (convc_kernels.hip; SFD2_PREC_F16C): hi plane + corr plane per tensor, see the top of this file
//   wpk: [2 * Cin / 32][ks * ks][Cout_pad][32] units (fp16 chunks, then corr chunks; the first half only when in_c is null)
//   in_c / res_c / out_c: corr planes (null = that tensor is plain fp16);  sbyte: the layer's E8M0 scale byte
void launch_convc_igemm(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int Cout_pad, int ks, int stride, int relu,
                        const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, int Ho, int Wo, int sbyte, unsigned int *range = nullptr);
// the tuned kernels' compensated instantiations (same wpk / planes / sbyte as launch_convc_igemm; null plane = plain fp16)
void launch_conv3x3_pp_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                         const float *scale, const float *shift, int CoutP, int relu, half_t *out, half_t *out_c,
                         int Ho, int Wo, const half_t *zero_page, int sbyte, const float *shift_sa6 = nullptr /* non-null: wpk's corr rows are fp6; [shift | scale bytes] */,
                         unsigned int *range = nullptr /* the output tensor's range-status slot (SFD2_RANGE_SUB words), here and below */,
                         int fmt6 = 0 /* bit 0: in_c holds fp6 half-records (then wpk / shift_sa6 are the fp6 x fp6 arrays), bit 1: out_c is written as fp6 half-records, bit 2 (with bit 0, without bit 1): the output is stored space-to-depth for conv2b_s2d_kernel, bit 3 (with bit 0, without bits 1 / 2): out_c holds the residual byte only (CoutP bytes per pixel) */);
bool conv3x3_rf_c_serves(int ks, int stride, int CoutP, int Cin, int Ho, int Wo);
// conv2b_s2d_kernel.hip: conv2b over conv2a's output stored space-to-depth (launch_conv3x3_pp_c with bit 2 of fmt6 writes that layout)
bool conv2b_s2d_serves(int H2, int W2, int Cin, int CoutP);
void launch_conv2b_s2d(hipStream_t st, const half_t *in_s2d, const half_t *in_c_s2d, int H4, int W4, const half_t *wpk, const float *scale,
                       const float *shift, int relu, half_t *out, half_t *out_c, const half_t *zero_page, int sbyte, unsigned int *range, int fmt6 /* bit 1: out_c as fp6 half-records */);
void launch_conv2b_s2d_x3(hipStream_t st, const half_t *in_hi_s2d, const half_t *in_lo_s2d, int H4, int W4, const half_t *wpk /* [4 hi | 4 lo' chunks][9][128][32] */,
                          const float *scale, const float *shift, int relu, half_t *out_hi, half_t *out_lo, const half_t *zero_page);
bool launch_conv3x3_rf_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                         const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out, half_t *out_c,
                         int Ho, int Wo, const half_t *zero_page, int sbyte, unsigned int *range = nullptr, int fmt6 = 0 /* bit 1: out_c as fp6 half-records */);
bool launch_conv_igemm2_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                          const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                          const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, int Ho, int Wo,
                          const half_t *zero_page, int sbyte, unsigned int *range = nullptr);
// compensated fused stem (fused_stem_c_kernel.hip): w1 = conv1a hi, lo fragments; w2 = conv1b register fragments
void launch_fused_stem_c(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *w1, const float *sc1,
                         const float *sh1, const void *w2, const float *sc2, const float *sh2, half_t *out, half_t *out_c,
                         int H2, int W2, int sbyte, unsigned int *range_base = nullptr /* the context's range-status words (conv1a's and conv1b's slots) */,
                         int fmt6 = 0 /* bit 1: out_c as fp6 half-records */);
void launch_conv1x1_c256_c(hipStream_t st, const half_t *in, const half_t *in_c, int npix, const half_t *w_frag,
                           const half_t *wc_frag, const float *scale, const float *shift, int relu, const half_t *res,
                           const half_t *res_c, half_t *out, half_t *out_c, const half_t *zero_page, int sbyte, unsigned int *range = nullptr,
                           int in_r1 = 0 /* in_c = residual bytes only (256 B per pixel): option "trunk_r1" */);
// sparse_da3_kernel.hip: convDa.3 on the four bilinear corner pixels of every selected key point only -> out [n_max][4][256] fp16
void launch_sparse_da3(hipStream_t st, const half_t *fmap, int hc, int wc, int nh, int nw, const half_t *wpk, const half_t *wsl, int CoutP,
                       const float *scale, const float *shift, int relu, const float *kpts, const unsigned int *count, int n_max,
                       half_t *out, const half_t *zero_page);
void launch_sparse_da3_repack(hipStream_t st, const half_t *w, half_t *dst, int CoutP, int cin);
void launch_sparse_da3_x3(hipStream_t st, const half_t *fmap_hi, const half_t *fmap_lo, int hc, int wc, int nh, int nw, const half_t *wpk, const half_t *wsl,
                          int CoutP, const float *scale, const float *shift, int relu, const float *kpts, const unsigned int *count,
                          int n_max, float *out, const half_t *zero_page);
// rb23_c_kernel.hip: ResBlock.conv2 + conv3 + residual in one kernel (SFD2_PREC_F16C, option "rb_inner" = 2: t1 plain fp16 in, t2 in LDS)
void launch_rb23_c(hipStream_t st, const half_t *t1, int H, int W, const half_t *w2h, const half_t *w2l, const float *sc2,
                   const float *sh2, const half_t *w3h, const half_t *w3l, const float *sc3, const float *sh3,
                   const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, const half_t *zero_page,
                   unsigned int *range_t2 = nullptr, unsigned int *range_out = nullptr,
                   int r1 = 0 /* residual-only corr bytes (256 B per pixel): bit 0 = res_c, bit 1 = out_c */,
                   const half_t *w3l8 = nullptr, int sbyte3 = 0 /* with r1 & 1: conv3's filter residuals as e4m3 in the kernel's K order, its corr scale byte */);
// conv1x1_kernels.hip: SFD2_PREC_F16X3 streaming 1x1 (256 -> 256): planes in, fp32 (+ planes) out, fp32 residual
void launch_conv1x1_c256_x3(hipStream_t st, const half_t *in, const half_t *in_lo, int npix, const half_t *w, const half_t *wl,
                            const float *scale, const float *shift, int relu, const void *res, const half_t *res_lo, float *out, half_t *out_hi,
                            half_t *out_lo, const half_t *zero_page);
void launch_conv1a_c(hipStream_t st, const float *img, int H, int W, int normalise, const half_t *wpk /*hi, lo fragments*/,
                     const float *scale, const float *shift, half_t *out, half_t *out_c, unsigned int *range = nullptr);
void launch_gconv_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, const half_t *wpk /*fp16 fragments*/,
                    const void *wck /*corr fragments*/, const float *scale, const float *shift, half_t *out, half_t *out_c, int sbyte,
                    int row0, int row1 /*output rows [row0, row1)*/, unsigned int *range = nullptr);
void launch_nhwc_hc_to_nchw_f(hipStream_t st, const half_t *in, const half_t *in_c, int npix, int pitch, int c, float *out, int fmt6 = 0 /* in_c: fp6 half-records */);

// ---- strict fp32 mode (conv_f32_kernels.hip): fp32 NHWC activations, f32-input MFMA
//   wpk [Cin/32][ks*ks][Cout_pad][32] fp32
void launch_conv_igemm_f32(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk,
                           const float *scale, const float *shift, int Cout_pad, int ks, int stride, int relu,
                           const float *residual, float *out, int Ho, int Wo);
// the same signature on the fp16 matrix path in three passes (SFD2_PREC_F16X3); wpk = the filters after launch_x3_split
void launch_x3_split(hipStream_t st, const float *w, size_t n_floats, void *out);
void launch_x3_split_planes(hipStream_t st, const float *in, size_t n_floats, void *hi_out, void *lo_out);
// conv3_kernels.hip: SFD2_PREC_F16X3 for the 3x3 stride-1 layers on pre-split planes (three passes of conv3x3_pp's fp16 K loop)
void launch_conv3x3_pp_x3(hipStream_t st, const half_t *in, const half_t *in_lo, int H, int W, int Cin, const half_t *wpk,
                          const float *scale, const float *shift, int CoutP, int relu, half_t *out_hi, half_t *out_lo, float *out_f32,
                          int Ho, int Wo, const half_t *zero_page, int s2d = 0 /* planes out stored space-to-depth (conv2b_s2d_kernel<x3>; Ho, Wo even) */);
void launch_gconv_x3_pack(hipStream_t st, const float *w /*[256][8][3][3]*/, void *out /*16 * 5 * 64 * 16 halves*/);
void launch_gconv_x3(hipStream_t st, const float *in, int H, int W, const void *wpk, const float *scale, const float *shift, float *out);
void launch_conv_igemm_x3(hipStream_t st, const float *in, int H, int W, int Cin, const float *wpk,
                           const float *scale, const float *shift, int Cout_pad, int ks, int stride, int relu,
                           const float *residual, float *out, int Ho, int Wo);
void launch_conv1a_f32(hipStream_t st, const float *img_chw, int H, int W, int normalise, const float *w /*[64][27]*/,
                       const float *scale, const float *shift, float *out /*[H][W][64]*/);
void launch_gconv_f32(hipStream_t st, const float *in, int H, int W, const float *w /*[256][72]*/, const float *scale,
                      const float *shift, float *out);
void launch_convsta_f32(hipStream_t st, const float *in, int npix, const float *w, const float *b, float *out);

// ------------------------------------------------------------------ heads / post
// logits [P][pitch] fp32 (65 used) -> score [8*hc][8*wc]
void launch_detector_head(hipStream_t st, const float *logits, int pitch, int hc, int wc, float *score);
// heat[H][W] = resize(score[hs][ws]) * stability(sta[3][hc][wc])   (sta may be null)
void launch_heatmap(hipStream_t st, const float *score, int hs, int ws, const float *sta, int hc, int wc,
                    int H, int W, float *heat, float *stab_out /*may be null*/);
// simple_nms + threshold + border; appends (score,idx) keys; optional dense output
// border test: x in [border, Wb-border), y in [border, Hb-border) -- Hb/Wb are the ORIGINAL image dims when the map
// is a rescaled pyramid level (nets/extractor.py:181-184 tests against W, H, not nw, nh)
// returns true when the top-K threshold search ran inside the NMS kernel (radius 4 with fuse_threshold)
bool launch_nms_select(hipStream_t st, const float *heat, int H, int W, int radius, float conf_th, int border, int Hb, int Wb,
                       float *nms_dense /*may be null*/, unsigned long long *cand_keys, int cand_cap,
                       unsigned int *counters /*[0]=n_cand*/, int fuse_threshold = 0, int top_k = 0);
// detector head + heat map in one kernel: logits [hc8 * wc8][pitch] (65 used), sta [3][hc][wc] or null -> heat [H][W];
// needs H == 8 * hc8 and W == 8 * wc8 (no score-map resize)
void launch_heads_heat(hipStream_t st, const float *logits, int pitch, int hc8, int wc8, const float *sta, int hc, int wc,
                       int H, int W, float *heat);
// top-K of the candidate keys, sorted descending -> sorted_keys[0..n_sel), counters[1]=n_sel
void launch_topk_sort(hipStream_t st, bool threshold_done, const unsigned long long *cand, int cand_cap, int top_k,
                      unsigned long long *sel, unsigned long long *sorted, int sel_cap, unsigned int *counters,
                      unsigned long long *bnd, int W, float *kpts, float *scores);   // kpts/scores: output rows (x, y), score
#define SFD2_HIST_BINS 4096                   // score bits >> 15, rebased to [2^-12, 2^4) and clamped (post_kernels.hip key_bin)
#define SFD2_COUNTER_BYTES (64 + SFD2_HIST_BINS * 4)   // 16 counters + score histogram
// histogram bin of a candidate key (score bits << 32 | ~index): the 16 exponent + mantissa bits below the sign, rebased
// so that bin 0 starts at 2^-12 and the last bin ends at 2^4, clamped.  Monotone in the score, which is all the
// threshold search needs: keys above the boundary bin are selected outright, the boundary bin is ranked exactly.
// (Heat-map scores are products of a soft-max probability and a stability weight: (0, 1].)
__device__ __forceinline__ unsigned int key_bin(unsigned long long key)
{
    const int b = (int)((unsigned int)(key >> 47) & 0xFFFFu) - ((127 - 12) << 8);
    return (unsigned int)(b < 0 ? 0 : (b > SFD2_HIST_BINS - 1 ? SFD2_HIST_BINS - 1 : b));
}

// threshold search over the 4096-bin score histogram by ONE block of 1024 threads (select_threshold_kernel, or the last
// block of nms4_select_kernel to finish).  wsum: 16 words of LDS.  COHERENT: the histogram and the candidate count were
// written by device-scope atomics of other blocks of THIS launch, so they are read with device-scope atomic loads (this
// part has one L2 per XCD; a release / acquire fence pair instead would write back and invalidate that L2 -- measured:
// 27 -> 83 us for the NMS kernel).
__device__ __forceinline__ unsigned int sfd2_ld_agent(const unsigned int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool COHERENT>
__device__ __forceinline__ void sfd2_select_threshold(int cand_cap, int top_k, unsigned int *__restrict__ counters, unsigned int *wsum)
{
    // 4096 bins = 1024 threads x 4 bins.  Per-thread group sums, suffix scan over the 1024 groups (wave shuffles +
    // 16 wave totals), then the thread owning the boundary group resolves the bin among its four.
    static_assert(SFD2_HIST_BINS == 4096, "one uint4 of bins per thread");
    const unsigned int *hist = counters + 16;
    unsigned int n = COHERENT ? sfd2_ld_agent(counters) : counters[0];
    if (n > (unsigned int)cand_cap) n = cand_cap;
    const unsigned int k = (top_k <= 0 || (unsigned int)top_k > n) ? n : (unsigned int)top_k;
    if (k == n) {
        if (threadIdx.x == 0) { counters[1] = n; counters[2] = 0; counters[3] = 0; counters[4] = 0; counters[5] = 0; counters[6] = 1; }
        return;
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint4 hv;
    if (COHERENT) hv = make_uint4(sfd2_ld_agent(hist + 4 * t), sfd2_ld_agent(hist + 4 * t + 1), sfd2_ld_agent(hist + 4 * t + 2), sfd2_ld_agent(hist + 4 * t + 3));
    else hv = reinterpret_cast<const uint4 *>(hist)[t];
    const unsigned int sum = hv.x + hv.y + hv.z + hv.w;
    unsigned int suf = sum;                    // inclusive suffix sum within the wave: groups t .. (wave end)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_down(suf, d);
        if (lane + d < 64) suf += o;
    }
    if (lane == 0) wsum[wave] = suf;
    __syncthreads();
    unsigned int higher = 0;                   // keys in all groups of higher waves
    for (int w = wave + 1; w < 16; ++w) higher += wsum[w];
    const unsigned int above_incl = higher + suf;          // keys in groups >= t
    unsigned int above = above_incl - sum;                 // keys in groups  > t
    if (above < k && above_incl >= k) {                    // exactly one thread: walk its bins from the top
        const unsigned int b[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (above < k && above + b[j] >= k) {
                counters[1] = k; counters[2] = 0; counters[3] = 0;
                counters[4] = 4 * t + j;       // boundary bin
                counters[5] = k - above;       // how many of its keys are selected
                counters[6] = 0;
            }
            above += b[j];
        }
    }
}

// nms4_kernels.hip: simple_nms with radius 4 + threshold + border + compaction (launch_nms_select serves other radii)
// fuse_threshold: the last block to finish also runs the top-K threshold search (counters[7] = its ticket counter);
// launch_topk_sort is then called with threshold_done
void launch_nms4_select(hipStream_t st, const float *heat, int H, int W, float conf_th, int border, int Hb, int Wb, float *nms_dense,
                        unsigned long long *cand, int cand_cap, unsigned int *counters, unsigned int *hist, int fuse_threshold, int top_k);
// greedy grid NMS of extract.py (nms_fast): init / one relaxation sweep / kept-score map
void launch_greedy_init(hipStream_t st, const float *heat, int n, float conf_th, unsigned long long *keys, unsigned char *state);
void launch_greedy_iter(hipStream_t st, const unsigned long long *keys, const unsigned char *sin, unsigned char *sout,
                        int H, int W, int dist, unsigned int *undecided);
void launch_greedy_final(hipStream_t st, const float *heat, const unsigned char *state, int n, float *kept);
// keys -> kpts (x,y), scores
void launch_keys_to_kpts(hipStream_t st, const unsigned long long *sorted_keys, const unsigned int *counters,
                         int W, float *kpts_xy, float *scores, int cap);
// bilinear sampling of desc_raw [hc][wc][128] fp32 NHWC at kpts + per-tap and final L2 normalisation
void launch_sample_desc(hipStream_t st, const float *desc_nhwc, int hc, int wc, int nh, int nw,
                        const float *kpts_xy, const unsigned int *count /*device, may be null*/, int n_max,
                        float *out, int compact = 0);
// sparse descriptor head (extract path): gather the key points' bilinear corner pixels, convDb on them, sample -- one kernel
void launch_pb_heads_heat(hipStream_t st, const half_t *fmap, int hc8, int wc8, const half_t *wpk, int CoutP, const float *scale,
                          const float *shift, const float *sta, int hc, int wc, int H, int W, float *heat,
                          unsigned int *zero_words /*nullable: n_zero words cleared if the grid covers them*/, int n_zero);
bool pb_heads_heat_clears(int hc8, int wc8, int n_zero);
void launch_desc_store64(hipStream_t st, const float *in /*[n_max][128]*/, const unsigned int *count, int n_max, double *out /*[128][pitch]*/, int pitch);
void launch_desc_head(hipStream_t st, const half_t *fmap, int hc, int wc, int nh, int nw, const half_t *wpk, int CoutP,
                      const float *scale, const float *shift, const float *kpts, const unsigned int *count, int n_max, float *out, int compact = 0);
// desc_raw NHWC [P][128] -> normalised NCHW [128][P]
void launch_desc_normalise_nchw(hipStream_t st, const float *desc_nhwc, int npix, float *out_nchw);
// layout helpers
void launch_nhwc_h_to_nchw_f(hipStream_t st, const half_t *in, int npix, int c_pitch, int c, float *out);
void launch_nhwc_f_to_nchw_f(hipStream_t st, const float *in, int npix, int c_pitch, int c, float *out);
void launch_nchw_f_to_nhwc_f(hipStream_t st, const float *in, int npix, int c, float *out);

void launch_scale_inplace(hipStream_t st, float *p, size_t n, float mul);
// *out = max(*out, max |in[i]|), as the bit pattern of the non-negative float
void launch_absmax_f32(hipStream_t st, const float *in, size_t n, unsigned int *out);

// decoder-side ingest: uint8 HWC (RGB or BGR) -> float32 CHW in [0,1] at nh x nw (cv2 INTER_CUBIC when the size changes)
void launch_ingest_u8(hipStream_t st, const unsigned char *src, int H, int W, int bgr, int nh, int nw, float *out);
void launch_unpack_rgbx(hipStream_t st, const unsigned char *src, unsigned char *dst, size_t npix);   // SFD2_FLAG_IMG_U8_X
void launch_extract_record(hipStream_t st, unsigned int *range_stat, unsigned int *hist, const unsigned int *counters, int sel_cap,
                           int cand_cap, unsigned int *rec /* 4 words, or null = fold the range words into hist only */);

// ------------------------------------------------------------------ matcher
// convert descriptors to fp16 [n][128] (hi) and optional scaled residual (lo)
void launch_match_prep(hipStream_t st, const void *src, int n, int n_src, const int *rows, int dim, int dtype, int layout,
                       half_t *hi, half_t *lo);
struct MatchJob {          // one direction of one pair
    const half_t *a_hi;    // reduced side ("columns" j), [na][128]
    const half_t *a_lo;
    const half_t *b_hi;    // kept side ("rows" i), [nb][128]
    const half_t *b_lo;
    int na, nb;
    float *part_v1;        // [splits][nb]
    float *part_v2;
    int *part_i1;
};
void launch_match_top2(hipStream_t st, const MatchJob *jobs_dev, int njobs, int max_nb, int splits, int use_lo,
                       int need_top2, const half_t *zero_page);
struct MatchJob2 {         // one pair, both directions from one GEMM (top-1 modes)
    const half_t *q_hi;    // queries [n0][128]
    const half_t *d_hi;    // database [n1][128]
    int n0, n1;
    float *part_v1;        // forward partials [splits][n0]
    int *part_i1;
    float *rkeys;          // reverse partials [ceil(n0/strip)][n1]: value with the strip-local query id in the low mantissa bits
};
struct MatchFinal;
// fwd_ids: the forward direction carries candidate ids (needed without the mutual check); 0 = mutual modes, the forward
// index is derived from the reverse direction (match_mutual_claim_kernel)
void launch_match_mutual(hipStream_t st, const MatchJob2 *jobs_dev, const MatchFinal *fins_dev, int npairs, int max_n0,
                         int max_n1, int splits, int fwd_ids);
int match_mutual_max_chunk(void);
int match_mutual_strip(void);        // queries per reverse-partial strip of the single-GEMM kernel   // most candidates one split of the single-GEMM kernel may sweep
void launch_match_decide(hipStream_t st, const MatchFinal *fin_dev, int npairs, int max_n, int flavour, int mutual,
                         float ratio, float dist);
struct MatchFinal {
    const float *f_v1, *f_v2; const int *f_i1;   // forward partials [splits][n0]
    const float *r_v1, *r_v2; const int *r_i1;   // reverse partials [splits][n1]
    int n0, n1;
    int out16, pad_;                             // SFD2_FLAG_MATCH_OUT16: matches0 / scores0 point to int16 / fp16 arrays (the stored types of hloc/match_features.py:114,118)
    long long *matches0; float *scores0;
    float *red_f;  // scratch [3*n0]
    float *red_r;  // scratch [3*n1]
    const int *remap;  // matched column -> caller's row index (gathered database sets), or null
};
void launch_match_finalize(hipStream_t st, const MatchFinal *fin_dev, int npairs, int max_n, int splits,
                           int flavour, int mutual, float ratio, float dist);

// persistent streaming 1x1 conv, 256 -> 256 (ResBlock conv1 / conv3); w_rowmajor = [256 out][256 in] fp16; zero page >= 512 B
void launch_conv1x1_c256(hipStream_t st, const half_t *in, int npix, const half_t *w_rowmajor, const float *scale,
                         const float *shift, int relu, const half_t *res, half_t *out, const half_t *zero_page);

// fused ResBlock: conv1 1x1 + grouped 3x3 + conv3 1x1 + residual (resblock_kernel.hip); w1 / w3 row-major [256][256] fp16,
// wg = the grouped conv's filters as compact [256 oc][9 taps][8 in] fp16; out must not alias x
void launch_resblock(hipStream_t st, const half_t *x, int H, int W, const half_t *w1, const float *sc1, const float *sh1,
                     const half_t *wg, const float *sc2, const float *sh2, const half_t *w3, const float *sc3, const float *sh3,
                     half_t *out, const half_t *zero_page);

// scale pyramid (nets/extractor.py:118-124,211-236,322-330)
void launch_norm_resize(hipStream_t st, const float *img, int mode, int H, int W, int nh, int nw, float *out);
void launch_ms_append(hipStream_t st, const float *kpts, const float *scores, const unsigned int *count, int cap, int W, int nw,
                      int H, int nh, float *kp_out, float *sc_out, unsigned int *level_count);
void launch_ms_merge(hipStream_t st, int n_levels, const int *offsets, const unsigned int *level_count, const float *kp_stage,
                     const float *sc_stage, const float *de_stage, int cap_total, int top_k, unsigned long long *keys,
                     unsigned long long *sorted, unsigned int *ms_counters, int n_max, float *kp_out, float *sc_out, float *de_out);
