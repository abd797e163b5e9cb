// conv_igemm2: second-generation implicit-GEMM conv for the stride-1 layers that carry most of
// the FLOPs (3x3 and 1x1, Cout tile 128 or 256).  Same GEMM orientation and epilogue as
// conv_kernels.hip; what changes is the staging:
//   * block = 512 threads = 8 waves, output tile 8 x 32 pixels x BN channels (the [BN][32] filter
//     tile of a K step is shared by twice as many pixels -> half the filter traffic per FLOP)
//   * global -> LDS copies are direct (global_load_lds_dwordx4): no VGPR round trip, no ds_write.
//     The LDS image is lane-linear (64-byte records), bank conflicts are avoided by an XOR swizzle of
//     the 16-byte slot index with ((record >> 2) & 3), applied to the per-lane SOURCE address when
//     staging and to the read address when forming fragments (same involution on both sides).
//   * out-of-image pixels (zero padding) read from a zero page instead of being predicated off,
//     because a masked lane would leave stale bytes in LDS.
// Replaces nets/sfd2.py Conv2d+BatchNorm2d+ReLU modules: conv2a, conv3a, conv3b (:272-278), the
// ResBlock 1x1 convs (:30-35), convPa.3 / convDa.0 / convDa.3 (:286-297), convPb / convDb (:299-300).
#include "sfd2_internal.h"
#include <stdlib.h>

#define TW 32
#define TH2 8
#define NT2 512

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __forceinline__ int xcd_swizzle2(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ h4_t cvt4b(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// COMP (SFD2_PREC_F16C, sfd2_internal.h; CC = 32 only): bit 0 = the input has a corr plane (in_c) and wpk holds
// 2 * Cin / 32 chunks -- the step loop runs on through the corr plane's chunks, whose units go to one
// v_mfma_scale_f32_32x32x64_f8f6f4 per (channel tile, pixel tile); bit 1 = the output's corr plane is written (out_c);
// the residual is hi + corr (res_c) when given.
template <int KS, int STRIDE, int BN, int CC, bool OUT_F32, bool HAS_RES, int NW = 8, int ROWS = 0, int XBUF = 2, int ABL = 0, int WBUF = 2, int TPS = 1, int COMP = 0>
__global__ __launch_bounds__(NW * 64, (XBUF == 1 && BN == 128 && STRIDE == 1) ? 4 : ((NW == 8 || ROWS != 0) ? 2 : 1))
void conv_igemm2_kernel(const half_t *__restrict__ in, int H, int W, int Cin,
                        const half_t *__restrict__ wpk, const float *__restrict__ scale,
                        const float *__restrict__ shift, int CoutP, int relu,
                        const half_t *__restrict__ res, void *__restrict__ outv,
                        int Ho, int Wo, int tiles_x, const half_t *__restrict__ zero_page,
                        const half_t *__restrict__ in_c = nullptr, const half_t *__restrict__ res_c = nullptr,
                        half_t *__restrict__ out_c = nullptr, int sa = 0, unsigned int *__restrict__ range = nullptr /* range-status slot of a compensated output */)
{
    static_assert(COMP == 0 || (CC == 32 && TPS == 1 && WBUF == 2 && XBUF != 3 && !OUT_F32 && ABL == 0), "compensated instantiations: 32-wide chunks, plain pipeline");
    constexpr int T = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int THT = ROWS ? ROWS : ((STRIDE == 1) ? TH2 : 4);   // output rows per tile: 8 (stride 1) or 4 (stride 2: the patch is 2x larger)
    constexpr int PH = (THT - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    constexpr int NPIX = PH * PW;
    constexpr int RB = CC * 2;                         // bytes per record (one pixel / one filter row of a K chunk)
    constexpr int RPC = 1024 / RB;                     // records per 1 KB LDS-DMA chunk (16 or 8)
    constexpr int SPR = RB / 16;                       // 16-byte slots per record (4 or 8)
    constexpr int SWS = (CC == 32) ? 2 : 1;            // swizzle: slot ^= (record >> SWS) & (SPR - 1)
    constexpr int XCH = (NPIX + RPC - 1) / RPC;        // 1 KB chunks of one patch
    constexpr int XPW = (XCH + NW - 1) / NW;           // chunks per wave
    // TPS = filter taps per pipeline stage (1, or 3 = one filter row: a third fewer barriers, 48 MFMAs per wave between them)
    static_assert(TPS == 1 || (TPS == 3 && KS == 3 && XBUF == 2 && WBUF == 2), "TPS = 3 is for the 3x3 two-buffer pipeline");
    constexpr int WCH = TPS * BN / RPC;                // 1 KB chunks of one stage's filter tiles
    constexpr int WPW = WCH / NW;                      // per wave
    constexpr int XBYTES = XCH * 1024, WBYTES = TPS * BN * RB;
    constexpr int WAVES_CH = 2, WAVES_PX = NW / 2;   // (4 x 2 and s_setprio around the MFMA bursts measured no better)
    constexpr int CH_T = BN / WAVES_CH / 32;           // 2 or 4
    constexpr int PX_T = THT / WAVES_PX;               // image rows per wave (2 or 1)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XBUF == 1: the input patch is single-buffered (its reload is exposed once per K chunk), which brings the
    // stride-2 tiles from 90 KB to 53 KB of LDS: three blocks per CU instead of one hide each other's step latencies
    unsigned char *Xs = smem;                          // [XBUF][XBYTES]
    unsigned char *Ws = smem + XBUF * XBYTES;          // [2][WBYTES]
    // XBUF == 3 (1x1 layers only): three input buffers, the chunk two steps ahead is already in flight, barriers wait
    // with a counted vmcnt.  With 256-channel tiles and 64-wide chunks that is exactly 160 KB, so scale/shift then
    // come from global memory in the epilogue instead of LDS.
    constexpr bool SS_LDS = !(XBUF == 3 && BN == 256 && CC == 64);
    // WBUF == 3 (3x3 layers): a ring of three filter tiles, the tile two steps ahead in flight across the barrier
    // (counted vmcnt), the next input chunk requested at the FIRST tap of the current one instead of the last.
    static_assert(WBUF == 2 || (WBUF == 3 && KS == 3 && XBUF == 2), "WBUF = 3 is the 3x3 filter ring");
    float *SS = reinterpret_cast<float *>(Ws + WBUF * WBYTES);   // scale[BN], shift[BN] of this channel tile (SS_LDS)
    static_assert(XBUF != 3 || (KS == 1 && STRIDE == 1), "XBUF = 3 is the 1x1 pipeline");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wch = (wave % WAVES_CH) * (CH_T * 32);
    const int wrow = (wave / WAVES_CH) * PX_T;

    const int n_tiles_n = CoutP / BN;
    const int swz = xcd_swizzle2(blockIdx.x, gridDim.x);
    const int tn = swz % n_tiles_n;
    const int tsp = swz / n_tiles_n;
    const int tx = tsp % tiles_x, ty = tsp / tiles_x;
    const int oy0 = ty * THT, ox0 = tx * TW, n0 = tn * BN;

    // ---- per-lane staging sources (element offsets into `in`; -1 = zero page)
    int xoff[XPW];
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        int chunk = wave + NW * i;
        if (WBUF == 3 && chunk >= XCH) chunk = XCH - 1;   // every wave issues XPW copies (the waits count them): duplicates of the last chunk
        const int q = chunk * RPC + lane / SPR;
        const int slot = (lane % SPR) ^ ((q >> SWS) & (SPR - 1));  // logical 16-byte slot this lane's bytes hold
        // stride 2: records are stored pair-swapped where bit 4 of the index is set, so that the B-fragment reads (lanes two
        // records apart) alternate between the halves of the bank space instead of all landing in one (2-way conflicts)
        const int ql = (STRIDE == 2) ? (q ^ ((q >> 4) & 1)) : q;   // logical record held at physical position q
        int off = -1;
        if (chunk < XCH && ql < NPIX) {
            const int py = ql / PW, px = ql - py * PW;
            const int iy = oy0 * STRIDE - PAD + py, ix = ox0 * STRIDE - PAD + px;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) off = (iy * W + ix) * Cin + slot * 8;
        }
        xoff[i] = off;
    }
    int woff[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int r = (wave * WPW + i) * RPC + lane / SPR;     // row of the stage tile: tap (r / BN), filter (r % BN)
        const int slot = (lane % SPR) ^ ((r >> SWS) & (SPR - 1));
        woff[i] = ((r / BN) * CoutP + n0 + (r % BN)) * CC + slot * 8;
    }

    const int NCHP = Cin / CC;                         // chunks per plane
#define ISSUE_X(chunk_, buf_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < XPW; ++i) {                                                  \
        if (ABL != 3 && !(ABL == 7 && (chunk_) != 0) && (WBUF == 3 || wave + NW * i < XCH)) {           \
            const int xc = (WBUF == 3 && wave + NW * i >= XCH) ? XCH - 1 : wave + NW * i;              \
            const half_t *pl_ = ((COMP & 1) && (chunk_) >= NCHP) ? in_c - (size_t)NCHP * CC : in;       \
            const half_t *src = xoff[i] >= 0 ? pl_ + (size_t)xoff[i] + (chunk_)*CC : zero_page + (lane % SPR) * 8; \
            __builtin_amdgcn_global_load_lds((gbl_void_t *)src,                                        \
                                             (lds_void_t *)(Xs + (buf_)*XBYTES + xc * 1024), 16, 0, 0); \
        }                                                                                              \
    }
#define ISSUE_W(step_, buf_)                                                                           \
    _Pragma("unroll") for (int i = 0; i < ((ABL == 4 || (ABL == 7 && (step_) != 0)) ? 0 : WPW); ++i) { \
        const half_t *src = wpk + (size_t)(step_)*TPS * CoutP * CC + woff[i];                          \
        __builtin_amdgcn_global_load_lds((gbl_void_t *)src,                                            \
                                         (lds_void_t *)(Ws + (buf_)*WBYTES + (wave * WPW + i) * 1024), 16, 0, 0); \
    }

    f32x16_t acc[CH_T][PX_T];
#pragma unroll
    for (int a = 0; a < CH_T; ++a)
#pragma unroll
        for (int b = 0; b < PX_T; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int NS = ((COMP & 1) ? 2 : 1) * (Cin / CC) * (T / TPS);   // pipeline stages (COMP: the hi plane's chunks, then the corr plane's)
    // barrier that lets the XPW most recently issued copies (the chunk two steps ahead) stay in flight
#define BARRIER_KEEP_X() asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(XPW) : "memory")
#define BARRIER_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
    ISSUE_X(0, 0)
    ISSUE_W(0, 0)
    if (SS_LDS)
        for (int t = tid; t < BN; t += NW * 64) { SS[t] = scale[n0 + t]; SS[BN + t] = shift[n0 + t]; }
#define BARRIER_KEEP(n_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(n_) : "memory")
    if (XBUF == 3) {
        if (NS > 1) { ISSUE_X(1, 1) BARRIER_KEEP_X(); } else { BARRIER_DRAIN(); }
    } else if (WBUF == 3) {
        if (NS > 1) { ISSUE_W(1, 1) BARRIER_KEEP(WPW); } else { BARRIER_DRAIN(); }
    } else {
        SFD2_BARRIER_DRAIN();   // hipcc drains vmcnt(0) before the barrier while LDS-DMA is in flight
    }

    const int lrow = lane & 31, lhi = lane >> 5;
    // fragment read offsets that do not depend on the step
    int a_off[CH_T];   // bytes: row * 64, plus the row's swizzle term kept separately
    int a_sw[CH_T];
#pragma unroll
    for (int ct = 0; ct < CH_T; ++ct) {
        const int r = wch + ct * 32 + lrow;
        a_off[ct] = r * RB;
        a_sw[ct] = (r >> SWS) & (SPR - 1);
    }

    int chunk = 0, tap = 0;
    for (int s = 0; s < NS; ++s) {
        const int wb = (WBUF == 3) ? (s % 3) : (s & 1), xb = (XBUF == 3) ? (chunk % 3) : ((XBUF == 2) ? (chunk & 1) : 0);
        int ntap = tap + TPS, nchunk = chunk;
        if (ntap == T) { ntap = 0; ++nchunk; }
        const bool has_next = (s + 1 < NS);
        const bool new_chunk = has_next && (ntap == 0);
        bool x_now = false, w_now = false;
        if (WBUF == 3) {
            x_now = (tap == 0) && (chunk + 1 < ((COMP & 1) ? 2 : 1) * (Cin / CC));
            w_now = s + 2 < NS;
            if (x_now) { ISSUE_X(chunk + 1, xb ^ 1) }        // nine steps ahead; older than every later filter copy
            if (w_now) { ISSUE_W(s + 2, (s + 2) % 3) }      // into the ring slot step s - 1 read
        } else {
            if (has_next) { ISSUE_W(s + 1, wb ^ 1) }
            if (XBUF == 2 && new_chunk) { ISSUE_X(nchunk, xb ^ 1) }
        }
        if (XBUF == 3 && s + 2 < NS) { ISSUE_X(s + 2, (s + 2) % 3) }   // after W(s + 1): the counted wait keeps exactly these

#pragma unroll
        for (int t3 = 0; t3 < TPS; ++t3) {
            const int ky = (tap + t3) / KS, kx = (tap + t3) - ky * KS;
            const unsigned char *xs = Xs + xb * XBYTES;
            const unsigned char *ws = Ws + wb * WBYTES + t3 * (BN * RB);
            int b_off[PX_T], b_sw[PX_T];
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr) {
                int q = ((wrow + pr) * STRIDE + ky) * PW + lrow * STRIDE + kx;
                if (STRIDE == 2) q ^= (q >> 4) & 1;                    // physical position of the record
                b_off[pr] = q * RB;
                b_sw[pr] = (q >> SWS) & (SPR - 1);
            }
            // software-pipelined fragment reads: the ds_reads of k-slice kk+1 are in flight while the
            // MFMAs of slice kk issue (two register sets, static indices)
            constexpr int NK = (ABL == 1) ? 0 : CC / 16;
            if ((COMP & 1) && chunk >= NCHP) {   // block-uniform: a corr chunk = both K slices of every fragment, one fp8 MFMA each
                v8i_t a8[CH_T], b8[PX_T];
#pragma unroll
                for (int ct = 0; ct < CH_T; ++ct)
                    a8[ct] = sfd2_cat8(*reinterpret_cast<const h8_t *>(ws + a_off[ct] + ((lhi ^ a_sw[ct]) << 4)),
                                       *reinterpret_cast<const h8_t *>(ws + a_off[ct] + (((2 + lhi) ^ a_sw[ct]) << 4)));
#pragma unroll
                for (int pr = 0; pr < PX_T; ++pr)
                    b8[pr] = sfd2_cat8(*reinterpret_cast<const h8_t *>(xs + b_off[pr] + ((lhi ^ b_sw[pr]) << 4)),
                                       *reinterpret_cast<const h8_t *>(xs + b_off[pr] + (((2 + lhi) ^ b_sw[pr]) << 4)));
#pragma unroll
                for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                    for (int pr = 0; pr < PX_T; ++pr)
                        acc[ct][pr] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[ct], b8[pr], acc[ct][pr], 0, 0, 0, sa, 0, 0x7f7f7f7f);
                // (the scaled MFMA is a pure node to instruction selection: without a use it may sink below the step's barrier)
#pragma unroll
                for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                    for (int pr = 0; pr < PX_T; ++pr) asm volatile("" : "+v"(acc[ct][pr]));
                continue;
            }
            h8_t fa[2][CH_T], fb[2][PX_T];
#pragma unroll
            for (int ct = 0; ct < CH_T; ++ct)
                fa[0][ct] = *reinterpret_cast<const h8_t *>(ws + a_off[ct] + ((lhi ^ a_sw[ct]) << 4));
#pragma unroll
            for (int pr = 0; pr < PX_T; ++pr)
                fb[0][pr] = *reinterpret_cast<const h8_t *>(xs + b_off[pr] + ((lhi ^ b_sw[pr]) << 4));
            // pin the issue order (hipcc otherwise sinks every read next to its first use and waits
            // lgkmcnt(0) in front of each MFMA group): reads of slice kk+1, then the MFMAs of slice kk
            __builtin_amdgcn_sched_group_barrier(0x100, CH_T + PX_T, 0);
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cur = kk & 1, nxt = cur ^ 1;
                if (kk + 1 < NK && ABL != 5) {   // ABL 5 (timing ablation): MFMAs on the first slice's fragments only
                    const int slot = (kk + 1) * 2 + lhi;
#pragma unroll
                    for (int ct = 0; ct < CH_T; ++ct)
                        fa[nxt][ct] = *reinterpret_cast<const h8_t *>(ws + a_off[ct] + ((slot ^ a_sw[ct]) << 4));
#pragma unroll
                    for (int pr = 0; pr < PX_T; ++pr)
                        fb[nxt][pr] = *reinterpret_cast<const h8_t *>(xs + b_off[pr] + ((slot ^ b_sw[pr]) << 4));
                }
#pragma unroll
                for (int ct = 0; ct < CH_T; ++ct)
#pragma unroll
                    for (int pr = 0; pr < PX_T; ++pr) {
                        if (ABL == 6) {   // timing ablation: fragment reads without the MFMAs
                            acc[ct][pr][0] += (float)fa[cur][ct][0] + (float)fb[cur][pr][0];
                        } else {
                            acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][ct], fb[cur][pr], acc[ct][pr], 0, 0, 0);
                        }
                    }
                if (kk + 1 < NK) __builtin_amdgcn_sched_group_barrier(0x100, CH_T + PX_T, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CH_T * PX_T, 0);
            }
        }
        if (XBUF == 3) {
            if (s + 2 < NS) BARRIER_KEEP_X(); else BARRIER_DRAIN();
        } else if (WBUF == 3) {
            // the filter tile of step s + 1 was issued one step ago; only what this step issued may stay in flight
            if (x_now && w_now) BARRIER_KEEP(XPW + WPW);
            else if (w_now) BARRIER_KEEP(WPW);
            else if (x_now) BARRIER_KEEP(XPW);
            else BARRIER_DRAIN();
        } else {
            SFD2_BARRIER_DRAIN();
        }
        if (XBUF == 1 && new_chunk) {
            ISSUE_X(nchunk, 0)
            SFD2_BARRIER_DRAIN();
        }
        tap = ntap;
        chunk = nchunk;
    }
#undef ISSUE_X
#undef ISSUE_W
#undef BARRIER_KEEP_X
#undef BARRIER_KEEP
#undef BARRIER_DRAIN

    // epilogue: y = acc * scale + shift (+ residual) (ReLU).  A lane owns, per register quad q, channels
    // 8q + 4*lhi .. +3 of one pixel, i.e. lanes l and l+32 hold the two 8-byte halves of one 16-byte
    // run.  For the fp16 output a v_permlane32_swap per dword regroups two quads so that lanes 0-31
    // own the full 16 bytes of quad 2m and lanes 32-63 those of quad 2m+1: half as many stores (and
    // residual loads), each 16 bytes wide.
    float mx = 0.0f;
#pragma unroll
    for (int pr = 0; pr < PX_T; ++pr) {
        const int oy = oy0 + wrow + pr, ox = ox0 + lrow;
        const bool inb = oy < Ho && ox < Wo;
        const size_t pix = (size_t)(inb ? oy : 0) * Wo + (inb ? ox : 0);
#pragma unroll
        for (int ct = 0; ct < CH_T; ++ct) {
            const int cl = wch + ct * 32 + 4 * lhi;        // channel within the tile, + 8q
            if (OUT_F32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 sc = SS_LDS ? *reinterpret_cast<const float4 *>(SS + cl + 8 * q) : *reinterpret_cast<const float4 *>(scale + n0 + cl + 8 * q);
                    const float4 sh = SS_LDS ? *reinterpret_cast<const float4 *>(SS + BN + cl + 8 * q) : *reinterpret_cast<const float4 *>(shift + n0 + cl + 8 * q);
                    float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                    float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                    float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                    float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                    if (relu) {
                        v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                    }
                    if (inb)
                        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(outv) + pix * CoutP + n0 + cl + 8 * q) =
                            make_float4(v0, v1, v2, v3);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    // 16-byte column run this lane ends up owning: quad 2m (lanes 0-31) or 2m+1 (lanes 32-63)
                    const size_t o16 = pix * CoutP + n0 + wch + ct * 32 + 8 * (2 * m + lhi);
                    uint2 rq[2] = {make_uint2(0, 0), make_uint2(0, 0)};
                    if (HAS_RES) {
                        uint4 r16 = make_uint4(0, 0, 0, 0);
                        if (inb) r16 = *reinterpret_cast<const uint4 *>(res + o16);
                        // undo the regrouping: afterwards rq[j] is this lane's 8-byte piece of quad 2m+j
                        const auto s0 = __builtin_amdgcn_permlane32_swap(r16.x, r16.z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(r16.y, r16.w, false, false);
                        rq[0] = make_uint2(s0[0], s1[0]);
                        rq[1] = make_uint2(s0[1], s1[1]);
                    }
                    uint2 rcq[2] = {make_uint2(0, 0), make_uint2(0, 0)};
                    if (HAS_RES && COMP) {   // the residual's corr units, regrouped like its hi plane
                        uint4 c16 = make_uint4(0, 0, 0, 0);
                        if (inb && res_c) c16 = *reinterpret_cast<const uint4 *>(res_c + o16);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(c16.x, c16.z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(c16.y, c16.w, false, false);
                        rcq[0] = make_uint2(s0[0], s1[0]);
                        rcq[1] = make_uint2(s0[1], s1[1]);
                    }
                    uint2 pk[2], ck[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int q = 2 * m + j;
                        const float4 sc = SS_LDS ? *reinterpret_cast<const float4 *>(SS + cl + 8 * q) : *reinterpret_cast<const float4 *>(scale + n0 + cl + 8 * q);
                        const float4 sh = SS_LDS ? *reinterpret_cast<const float4 *>(SS + BN + cl + 8 * q) : *reinterpret_cast<const float4 *>(shift + n0 + cl + 8 * q);
                        if (COMP & 2) {   // compensated output: packed-fp32 epilogue, one med3 per value (sfd2_internal.h)
                            float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (HAS_RES) {
                                h4_t r;
                                __builtin_memcpy(&r, &rq[j], 8);
                                ad = make_float4((float)r[0] + sfd2_corr_lo(rcq[j].x, 0), (float)r[1] + sfd2_corr_lo(rcq[j].x, 1),
                                                 (float)r[2] + sfd2_corr_lo(rcq[j].y, 0), (float)r[3] + sfd2_corr_lo(rcq[j].y, 1));
                            }
                            sfd2_epi4<HAS_RES>(acc[ct][pr][4 * q + 0], acc[ct][pr][4 * q + 1], acc[ct][pr][4 * q + 2], acc[ct][pr][4 * q + 3], sc, sh, ad,
                                               relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j], mx, inb);
                            continue;
                        }
                        float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                        float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                        float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                        float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                        if (HAS_RES) {
                            h4_t r;
                            __builtin_memcpy(&r, &rq[j], 8);
                            v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
                            if (COMP) {
                                v0 += sfd2_corr_lo(rcq[j].x, 0); v1 += sfd2_corr_lo(rcq[j].x, 1);
                                v2 += sfd2_corr_lo(rcq[j].y, 0); v3 += sfd2_corr_lo(rcq[j].y, 1);
                            }
                        }
                        if (relu) {
                            v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f);
                        }
                        if (COMP & 2) {
                            sfd2_split4(v0, v1, v2, v3, pk[j], ck[j]);
                        } else {
                            const h4_t hv = cvt4b(v0, v1, v2, v3);
                            __builtin_memcpy(&pk[j], &hv, 8);
                        }
                    }
                    const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                    if (inb && (ABL != 2 || t0[0] == 0x12345678u))
                        *reinterpret_cast<uint4 *>(reinterpret_cast<half_t *>(outv) + o16) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                    if (COMP & 2) {
                        const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                        const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                        if (inb) *reinterpret_cast<uint4 *>(out_c + o16) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                    }
                }
            }
        }
    }
    if (COMP & 2) sfd2_range_commit(range, sfd2_wave_max_bits(mx));
}

template <int KS, int STRIDE, int BN, int CC, bool OUT_F32, bool HAS_RES, int NW = 8, int ROWS = 0, int XBUF = 2, int ABL = 0, int WBUF = 2, int TPS = 1, int COMP = 0>
static void launch_igemm2_t(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                            const float *scale, const float *shift, int CoutP, int relu, const half_t *res,
                            void *out, int Ho, int Wo, const half_t *zero_page, const half_t *in_c = nullptr,
                            const half_t *res_c = nullptr, half_t *out_c = nullptr, int sa = 0, unsigned int *range = nullptr)
{
    constexpr int THT = ROWS ? ROWS : ((STRIDE == 1) ? TH2 : 4);
    constexpr int PH = (THT - 1) * STRIDE + KS, PW = (TW - 1) * STRIDE + KS;
    constexpr int RPC = 1024 / (CC * 2);
    constexpr int XCH = (PH * PW + RPC - 1) / RPC;
    constexpr bool SS_LDS = !(XBUF == 3 && BN == 256 && CC == 64);
    constexpr size_t lds = (size_t)XBUF * XCH * 1024 + (size_t)WBUF * TPS * BN * CC * 2 + (SS_LDS ? (size_t)2 * BN * sizeof(float) : 0);
    static bool attr_done = false;
    auto kern = conv_igemm2_kernel<KS, STRIDE, BN, CC, OUT_F32, HAS_RES, NW, ROWS, XBUF, ABL, WBUF, TPS, COMP>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + THT - 1) / THT;
    const int grid = tiles_x * tiles_y * (CoutP / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, res, out,
                       Ho, Wo, tiles_x, zero_page, in_c, res_c, out_c, sa, range);
}

// compensated instantiations (SFD2_PREC_F16C; in and out compensated): 1x1 256 -> 256 (+ residual) of the ResBlocks and the
// stride-2 3x3 layers with 128 / 256 output channels.  wpk = the layer's wc array (32-wide chunks).  false = no instantiation.
bool launch_conv_igemm2_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                          const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                          const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, int Ho, int Wo,
                          const half_t *zero_page, int sbyte, unsigned int *range)
{
    const int sa = (sbyte & 255) * 0x01010101;
    if (!in_c || !out_c) return false;
    if (ks == 1 && stride == 1 && CoutP % 128 == 0) {   // 128-channel tiles: the 256-channel tile's compensated epilogue spills
        if (res) launch_igemm2_t<1, 1, 128, 32, false, true, 8, 0, 2, 0, 2, 1, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, res, out, Ho, Wo, zero_page, in_c, res_c, out_c, sa, range);
        else launch_igemm2_t<1, 1, 128, 32, false, false, 8, 0, 2, 0, 2, 1, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page, in_c, nullptr, out_c, sa, range);
        return true;
    }
    if (ks == 3 && stride == 2 && !res && CoutP % 128 == 0) {
        if (CoutP % 256 == 0) launch_igemm2_t<3, 2, 256, 32, false, false, 8, 0, 1, 0, 2, 1, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page, in_c, nullptr, out_c, sa, range);
        else launch_igemm2_t<3, 2, 128, 32, false, false, 8, 0, 1, 0, 2, 1, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page, in_c, nullptr, out_c, sa, range);
        return true;
    }
    return false;
}

bool conv3x3_pp_serves(int ks, int stride, int CoutP, int Cin);
void launch_conv3x3_pp(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                       const float *scale, const float *shift, int CoutP, int relu, half_t *out,
                       int Ho, int Wo, const half_t *zero_page);

bool conv3x3_rf_serves(int ks, int stride, int CoutP, int Cin, int Ho, int Wo);
bool conv3x3_rf_resident(int ks, int stride, int CoutP, int Cin);
void launch_conv3x3_rf(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                       const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out,
                       int Ho, int Wo, const half_t *zero_page);

// K-chunk width the v2 kernel wants the filters packed with (0 = layer is not served by v2)
int conv_igemm2_chunk(int ks, int stride, int CoutP, int Cin)
{
    if (CoutP % 128 != 0) return 0;
    if (conv3x3_pp_serves(ks, stride, CoutP, Cin)) return 32;   // conv3_kernels.hip
    if (conv3x3_rf_resident(ks, stride, CoutP, Cin)) return 32; // conv3rf_kernels.hip (conv2a)
    if (stride == 2) return (ks == 3) ? 32 : 0;        // stride-2 3x3: 4-row tiles, 32-wide chunks (patch 9 x 65 records)
    if (stride != 1) return 0;
    // 256-channel tiles: 64-wide K chunks (one block per CU either way, half the barriers);
    // 128-channel tiles: 32-wide chunks keep the LDS footprint at 60 KB -> two blocks per CU
    static const bool bn128 = sfd2_env("SFD2_CONV_BN128") != nullptr;   // experiment: 128-channel tiles everywhere
    static const bool small1 = sfd2_env("SFD2_CONV_1X1_SMALL") != nullptr;   // experiment: 4-wave 4x32 tiles for 1x1
    if (small1 && ks == 1 && CoutP % 256 == 0) return 32;
    static const bool w3 = sfd2_env("SFD2_CONV_WRING") != nullptr || sfd2_env("SFD2_CONV_TPS3") != nullptr;   // experiments on 32-wide chunks
    if (w3 && ks == 3 && stride == 1 && CoutP % 256 == 0) return 32;
    if (CoutP % 256 == 0 && !bn128) return (Cin % 64 == 0) ? 64 : 32;
    // conv2a (64 -> 128): the whole K of a tap row fits one 64-wide chunk, so the patch is staged once (XBUF = 1,
    // 76 KB of LDS, two blocks per CU) and the 9 steps carry 16 MFMAs each instead of 18 steps of 8
    static const bool c2a32 = sfd2_env("SFD2_CONV2A_CC32") != nullptr;
    if (ks == 3 && Cin == 64 && CoutP == 128 && !c2a32) return 64;
    return 32;
}

// returns false when the (ks, Cout tile) combination has no v2 instantiation
bool launch_conv_igemm2(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int CoutP, int ks, int stride, int relu,
                        const half_t *residual, void *out, int out_f32, int Ho, int Wo, const half_t *zero_page)
{
    if (!residual && !out_f32 && conv3x3_rf_serves(ks, stride, CoutP, Cin, Ho, Wo)) {   // conv3rf_kernels.hip: small outputs
        launch_conv3x3_rf(st, in, H, W, Cin, wpk, scale, shift, CoutP, stride, relu, reinterpret_cast<half_t *>(out), Ho, Wo, zero_page);
        return true;
    }
    if (stride == 2) {
        if (conv_igemm2_chunk(ks, 2, CoutP, Cin) != 32 || residual || out_f32) return false;
        // single-buffered input patch (XBUF = 1): 53 / 69 KB of LDS instead of 90 / 106 KB, so 3 / 2 blocks share a CU
        // and hide each other's per-step latencies (steps are only 8-16 MFMAs long here).  Measured at 1600x1200:
        // conv2b 98 -> 70 us, convPa.0 120 -> 82 us.  SFD2_CONV_S2_XBUF2 restores the double-buffered variant.
#ifdef SFD2_EXPERIMENTS
        static const bool xb2 = sfd2_env("SFD2_CONV_S2_XBUF2") != nullptr;
        if (xb2) {
            if (CoutP % 256 == 0) launch_igemm2_t<3, 2, 256, 32, false, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page);
            else launch_igemm2_t<3, 2, 128, 32, false, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page);
            return true;
        }
#endif
        if (CoutP % 256 == 0) launch_igemm2_t<3, 2, 256, 32, false, false, 8, 0, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page);
        else launch_igemm2_t<3, 2, 128, 32, false, false, 8, 0, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, nullptr, out, Ho, Wo, zero_page);
        return true;
    }
#define SFD2_IG2B(KS_, BN_, CC_, F32_)                                                                                   \
    do {                                                                                                                \
        if (residual) launch_igemm2_t<KS_, 1, BN_, CC_, F32_, true>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); \
        else launch_igemm2_t<KS_, 1, BN_, CC_, F32_, false>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);         \
    } while (0)
#define SFD2_IG2(KS_, BN_, F32_)                                         \
    do {                                                                 \
        if (cc == 64) SFD2_IG2B(KS_, BN_, 64, F32_);                     \
        else SFD2_IG2B(KS_, BN_, 32, F32_);                              \
    } while (0)
    const int cc = conv_igemm2_chunk(ks, 1, CoutP, Cin);
    if (cc == 0) return false;
    if (conv3x3_pp_serves(ks, 1, CoutP, Cin) && !residual && !out_f32) {
        launch_conv3x3_pp(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, reinterpret_cast<half_t *>(out), Ho, Wo, zero_page);
        return true;
    }
    static const bool bn128 = sfd2_env("SFD2_CONV_BN128") != nullptr;
    const int bn = (CoutP % 256 == 0 && !bn128) ? 256 : 128;
#ifdef SFD2_EXPERIMENTS
    static const bool nw4 = sfd2_env("SFD2_CONV_NW4") != nullptr;   // experiment: 4 waves, 128 ch x 128 px per wave
    if (nw4 && ks == 3 && !out_f32 && bn == 256 && cc == 64 && !residual) {
        launch_igemm2_t<3, 1, 256, 64, false, false, 4>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        return true;
    }
    static const bool small1 = sfd2_env("SFD2_CONV_1X1_SMALL") != nullptr;
    if (small1 && ks == 1 && !out_f32 && bn == 256 && cc == 32) {
        if (residual) launch_igemm2_t<1, 1, 256, 32, false, true, 4, 4>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        else launch_igemm2_t<1, 1, 256, 32, false, false, 4, 4>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        return true;
    }
#endif
#ifdef SFD2_EXPERIMENTS
    if (const char *ab = sfd2_env("SFD2_CONV_3X3_ABLATE")) {   // timing ablations of the dominant kernel (wrong results)
        if (ks == 3 && !out_f32 && bn == 256 && cc == 64 && !residual) {
            if (ab[0] == '5') { launch_igemm2_t<3, 1, 256, 64, false, false, 8, 0, 2, 5>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true; }
            if (ab[0] == '6') { launch_igemm2_t<3, 1, 256, 64, false, false, 8, 0, 2, 6>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true; }
            if (ab[0] == '7') { launch_igemm2_t<3, 1, 256, 64, false, false, 8, 0, 2, 7>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true; }
            if (ab[0] == '1') { launch_igemm2_t<3, 1, 256, 64, false, false, 8, 0, 2, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true; }
        }
    }
#endif
#ifdef SFD2_EXPERIMENTS
    static const bool tps3 = sfd2_env("SFD2_CONV_TPS3") != nullptr;   // experiment: one filter ROW (3 taps) per pipeline stage
    if (tps3 && ks == 3 && !out_f32 && bn == 256 && cc == 32 && !residual) {
        launch_igemm2_t<3, 1, 256, 32, false, false, 8, 0, 2, 0, 2, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        return true;
    }
    static const bool w3 = sfd2_env("SFD2_CONV_WRING") != nullptr;
    if (w3 && ks == 3 && !out_f32 && bn == 256 && cc == 32 && !residual) {
        launch_igemm2_t<3, 1, 256, 32, false, false, 8, 0, 2, 0, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        return true;
    }
#endif
    if (ks == 3 && !out_f32 && bn == 128 && cc == 64 && Cin == 64 && !residual) {
        launch_igemm2_t<3, 1, 128, 64, false, false, 8, 0, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);
        return true;
    }
    if (ks == 3 && !out_f32) { if (bn == 256) SFD2_IG2(3, 256, false); else SFD2_IG2(3, 128, false); return true; }
    // Experiment (SFD2_CONV_1X1_XBUF3): three input buffers with the chunk two steps ahead in flight and counted-vmcnt
    // barriers for the 1x1 layers.  Correct, but measured SLOWER than the two-buffer pipeline (conv1 34 -> 39 us,
    // conv3 44.5 -> 49 us at 1600x1200): doubling the input bytes in flight is not what these layers lack.
#ifdef SFD2_EXPERIMENTS
    if (const char *ab = sfd2_env("SFD2_CONV_1X1_ABLATE")) {   // timing ablations of the 1x1 256-channel layer (wrong results)
        if (ks == 1 && !out_f32 && bn == 256 && cc == 64 && !residual) {
            switch (ab[0]) {
            case '1': launch_igemm2_t<1, 1, 256, 64, false, false, 8, 0, 2, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true;
            case '2': launch_igemm2_t<1, 1, 256, 64, false, false, 8, 0, 2, 2>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true;
            case '3': launch_igemm2_t<1, 1, 256, 64, false, false, 8, 0, 2, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true;
            case '4': launch_igemm2_t<1, 1, 256, 64, false, false, 8, 0, 2, 4>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); return true;
            default: break;
            }
        }
    }
#endif
#ifdef SFD2_EXPERIMENTS
    static const bool x3 = sfd2_env("SFD2_CONV_1X1_XBUF3") != nullptr;
#define SFD2_IG1(BN_, CC_, F32_)                                                                                         \
    do {                                                                                                                \
        if (residual) launch_igemm2_t<1, 1, BN_, CC_, F32_, true, 8, 0, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page); \
        else launch_igemm2_t<1, 1, BN_, CC_, F32_, false, 8, 0, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, residual, out, Ho, Wo, zero_page);         \
    } while (0)
    if (ks == 1 && x3 && !out_f32 && bn == 256 && cc == 64) { SFD2_IG1(256, 64, false); return true; }
    if (ks == 1 && x3 && out_f32 && bn == 128 && cc == 32) { SFD2_IG1(128, 32, true); return true; }
#undef SFD2_IG1
#endif
    if (ks == 1 && !out_f32) { if (bn == 256) SFD2_IG2(1, 256, false); else SFD2_IG2(1, 128, false); return true; }
    if (ks == 1 && out_f32) { if (bn == 256) SFD2_IG2(1, 256, true); else SFD2_IG2(1, 128, true); return true; }
#undef SFD2_IG2
#undef SFD2_IG2B
    return false;
}
