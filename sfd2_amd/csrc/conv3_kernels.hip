// conv3x3_pp: third-generation implicit-GEMM kernel for the 3x3 stride-1 layers with >= 256 output
// channels (conv3a / conv3b, convPa.3, convDa.0 / convDa.3: nets/sfd2.py:274-297), which carry ~60 % of
// the network's FLOPs.  Same GEMM orientation, LDS record layout / swizzle and epilogue as
// conv2_kernels.hip; what changes is the tile and the schedule.
//
//   * tile = 128 channels x (16 x 32) pixels, K chunk of 32 input channels.  The copies into LDS go
//     through the LDS-DMA path, which moves ~18 B/clk/CU -- with 256-channel x 256-pixel tiles the
//     37 KB a K step needs take as long as the step's MFMAs (75 us of copies alone on the 141-GFLOP
//     layers).  This shape needs two thirds of those bytes per FLOP (8 KB of filters per tap instead
//     of 2 x 8, the larger patch amortised over nine taps).
//   * the eight waves form two groups of four (one wave of each group on every SIMD) that run the
//     SAME code one barrier apart.  A unit of work = one filter tap x 32 channels = 12 fragment reads
//     + 16 MFMAs per wave, written as a LOAD section (fragment ds_reads, plus this wave's share of the
//     next staging copies) and an MFMA section, each closed by a workgroup barrier.  Because group 1
//     executed one extra barrier at the start, its LOAD section runs while group 0's MFMA section owns
//     the matrix pipe and vice versa: copies, LDS reads and address arithmetic of one wave are hidden
//     behind the other wave's MFMAs instead of both waves of a SIMD stalling together.
//
// Time in "slots" (the interval between two consecutive barriers); unit u = 9 * chunk + tap:
//     group 0:  LOAD(u) in slot 2u,      MFMA(u) in slot 2u + 1
//     group 1:  LOAD(u) in slot 2u + 1,  MFMA(u) in slot 2u + 2
// Staging (stage = one filter row ky of a chunk = 3 units, two buffers; patch = one chunk = 9 units,
// two buffers):
//     * filters of stage t + 1 are requested in LOAD(3t) (by each wave, for its own 3 pieces) into the
//       buffer stage t - 1 used: its last reader was group 1's LOAD(3t - 1), one slot before group 0's
//       LOAD(3t), so every LOAD of a stage's last tap retires its reads (lgkmcnt(0)) before its barrier.
//     * the patch of chunk c + 1 is requested one piece at a time in LOAD of taps 0, 1, 3, 4, 6 of
//       chunk c (same argument for the buffer's previous readers: tap 8 is a stage's last tap).
//     * both sections of a stage's last tap end with vmcnt(0): everything requested so far has landed
//       before the barrier that precedes the first read of the next stage (the youngest request is one
//       unit = two slots old at that point, so the wait is normally free).
#include "sfd2_internal.h"

#define PP_TW 32
#define PP_TH 16
#define PP_BN 128
#define PP_CC 32
#define PP_PW (PP_TW + 2)
#define PP_PH (PP_TH + 2)
#define PP_NPIX (PP_PH * PP_PW)                 // 612 patch records of 64 B
#define PP_XCH ((PP_NPIX + 15) / 16)            // 39 pieces of 1 KB
#define PP_XPW 5                                // pieces per wave (waves issue 5 each; the tail repeats the last piece)
#define PP_XBYTES (PP_XCH * 1024)
#define PP_FBYTES (3 * PP_BN * PP_CC * 2)       // one filter row: 3 taps x 128 filters x 64 B = 24 KB
#define PP_FPW 3                                // filter pieces per wave per stage (24 / 8)

typedef __attribute__((address_space(3))) void lds_void3_t;
typedef const __attribute__((address_space(1))) void gbl_void3_t;

__device__ __forceinline__ int xcd_swizzle3(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ h4_t cvt4c(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// STAGGER = 0 builds the same loop without the one-barrier offset (A/B switch for the schedule itself)
// experiment builds (timing only, wrong results): -DSFD2_PP_NO_CORR_MFMA drops the scaled MFMAs of the corr chunks, everything else stays
#ifdef SFD2_PP_NO_CORR_MFMA
#define PP_NO_CORR_MFMA 1
#else
#define PP_NO_CORR_MFMA 0
#endif
// -DSFD2_PP_ABL_BARRIER (timing only, WRONG results): the corr chunks skip both barriers of every stage's middle unit -- how much of a corr
// chunk's time is the per-slot synchronisation rather than its MFMAs, LDS reads or copies?
#ifdef SFD2_PP_ABL_BARRIER
#define PP_ABL_BARRIER 1
#else
#define PP_ABL_BARRIER 0
#endif
#ifndef SFD2_PP_KXM
#define SFD2_PP_KXM 1
#endif
// SFD2_PP_RING3 (compensated instantiations, not X3): the K loop takes the chunks INTERLEAVED -- hi chunk of channels 32 k, then the corr
// chunk of the same channels -- and the filter stages go through a ring of THREE buffers, requested two stages ahead, with counted waits
// that leave the newest requests in flight.  Why: a hi chunk is bound by its MFMAs (9 units x 16 x 32 cycles, two waves per SIMD: 9.2 k
// cycles against 6.3 k for its 111 KB of operand copies at ~18 B/clk/CU), a corr chunk by its copies (4.8 k of MFMA against the same
// 6.3 k).  Run as two blocks of eight the second block waits for its copies throughout; interleaved, with two stages of lookahead,
// the copies average over both kinds (222 KB per 14 k cycles of MFMA issue).
#ifndef SFD2_PP_RING3
#define SFD2_PP_RING3 0
#endif
// KXM = 1: the nine taps of a chunk run column by column (stage = one filter COLUMN kx, units ky = 0, 1, 2), so that the
// pixel fragments of patch row r serve output rows r, r - 1, r - 2 of the same stage from registers: a stage reads six
// patch rows once (4 + 1 + 1 over its three units) instead of four rows per unit -- 8 fragment reads per unit instead of 12.
// COMP (SFD2_PREC_F16C, sfd2_internal.h): bit 0 = the input has a corr plane (in_c) and wpk holds 2 * Cin / 32 chunks --
// the chunk loop simply runs on through the corr plane's chunks, whose units go to ONE v_mfma_scale_f32_32x32x64_f8f6f4 per
// (channel tile, pixel tile) instead of two fp16 MFMAs (same LDS records, same fragment reads); bit 1 = the corr plane of
// the output is written (out_c); bit 4 = the corr FILTER rows are fp6 (e2m3) strings with one E8M0 scale byte per output channel
// (behind the shifts: shift[CoutP + channel] holds the byte replicated into an int; api_weights.hip pack_igemm): the same fragment reads (the strings sit in the 16-byte slots the lane halves read), the MFMA
// runs fp8 x fp6 -- 32 ns per instruction and SIMD instead of 37-41 (profiles/r03k_mfma_f8f6f4_probe.txt).
// Measured with the trace below (conv2a compensated, 64 -> 128 channels: K loop 45k cycles, epilogue 15-22k, wait at the next
// tile's top 2-9k): the epilogue is long because every CU writes its 256 KB tile at the same time (64 MB per round, the K loops
// being equally long everywhere), not because of the wait the compiler puts in front of the epilogue's first LDS read (it orders
// LDS accesses behind pending direct-to-LDS copies it cannot prove disjoint, here the next tile's first copies): with the copies
// issued by inline asm -- no such wait -- and the next tile's scale / shift stored before them, conv2a 147.6 -> 152.9 us,
// conv3a 119.3 -> 121.0, conv3b and convDa unchanged.  Not kept.  Non-temporal epilogue stores (so that the burst does not evict the
// patches and filters from the L2): conv2a 150 -> 299 us, conv3a 123 -> 196, conv3b 216 -> 278 -- the write-back L2 is what absorbs
// the burst.
// -DSFD2_PP_TRACE: wall-clock stamps (s_memrealtime, 100 MHz) of every block (entry, exit) and cycle stamps of block 3's waves 0
// and 4 around the sections of its tiles, printed by the launcher for the compensated instantiations
#ifdef SFD2_PP_TRACE
#include <stdio.h>
__device__ unsigned long long g_pp_wall[1024][2];
__device__ unsigned long long g_pp_cyc[2][8][4];
#define PP_WALL(k_) if (tid == 0) g_pp_wall[blockIdx.x][k_] = __builtin_amdgcn_s_memrealtime();
#define PP_CYC(k_) if (blockIdx.x == 3 && (wave == 0 || wave == 4) && lane == 0 && it < 8) g_pp_cyc[wave == 4][it][k_] = __builtin_readcyclecounter();
// unit-level stamps of the second tile's chunk 1 (a hi chunk) and chunk NCH + 1 (a corr chunk): LOAD begin, LOAD end (in front of its barrier), MFMA
// begin (behind it), MFMA end (in front of the second barrier)
__device__ unsigned long long g_pp_unit[2][2][9][4];
#define PP_UCYC(k_) if (blockIdx.x == 3 && it == 1 && (c == 1 || c == NCH + 1) && (wave == 0 || wave == 4) && lane == 0) \
        g_pp_unit[wave == 4][c != 1][t9][k_] = __builtin_readcyclecounter();
#else
#define PP_WALL(k_)
#define PP_CYC(k_)
#define PP_UCYC(k_)
#endif

template <int STAGGER, int PRIO, int ABL = 0, int KXM = SFD2_PP_KXM, int COMP = 0>
__global__ __launch_bounds__(512, 2)
void conv3x3_pp_kernel(const half_t *__restrict__ in, int H, int W, int Cin,
                       const half_t *__restrict__ wpk, const float *__restrict__ scale,
                       const float *__restrict__ shift, int CoutP, int relu,
                       half_t *__restrict__ out, int Ho, int Wo, int tiles_x, int n_tiles,
                       const half_t *__restrict__ zero_page,
                       const half_t *__restrict__ in_c = nullptr, half_t *__restrict__ out_c = nullptr, int sa = 0,
                       unsigned int *__restrict__ range = nullptr /* the output tensor's range-status slot (compensated output) */)
{
    constexpr bool F6 = (COMP & 16) != 0;
    constexpr bool B6 = (COMP & 32) != 0;                  // the INPUT's corr records are fp6 half-records (sfd2_internal.h; with F6: fp6 x fp6, 33.5 cycles per scaled MFMA)
    constexpr bool O6 = (COMP & 64) != 0;                  // the OUTPUT's corr records are written as fp6 half-records
    constexpr bool OR1 = (COMP & 256) != 0;                // the OUTPUT's corr plane holds the residual byte only, CoutP bytes per pixel (option "trunk_r1": conv3b -> ResBlock 0)
    constexpr bool S2D = (COMP & 128) != 0;                // the OUTPUT is stored space-to-depth: [Ho / 2][Wo / 2][(y & 1) * 2 + (x & 1)][CoutP] (conv2b_s2d_kernel.hip; Ho, Wo even)
    static_assert(!B6 || F6, "fp6 pixel operands come with fp6 filter strings");
    constexpr int SSN = F6 ? 3 : 2;                        // arrays per tile parity in SSb: scale, shift (, the fp6 filters' scale bytes)
    constexpr bool RING3 = (SFD2_PP_RING3 != 0) && (COMP & 1) && !(COMP & 4) && KXM && !ABL;   // three filter buffers, counted waits (above)
    constexpr bool ILV = RING3 && (SFD2_PP_RING3 & 1);      // bit 0: hi and corr chunks interleaved
    constexpr bool SPREAD = RING3 && (SFD2_PP_RING3 & 2);   // bit 1: a stage's three filter pieces are requested one per unit instead of all in its first unit
    constexpr int NFB = RING3 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                              // [2][PP_XBYTES]
    unsigned char *Fs = smem + 2 * PP_XBYTES;              // [NFB][PP_FBYTES]
    float *SSb = reinterpret_cast<float *>(Fs + NFB * PP_FBYTES);  // [2] x (scale[128], shift[128]): one set per tile parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                             // waves w and w + 4 share a SIMD
    const int wch = (wave & 1) * 64;                       // 2 channel tiles
    const int wrow = (wave >> 1) * 4;                      // 4 image rows = 4 pixel tiles

    // PERSISTENT blocks (one per CU: 127 KB of LDS): tiles blockIdx.x, + gridDim.x, ...  The next tile's first patch and
    // filter row are requested BEFORE the current tile's epilogue, so they land behind its conversions and stores instead
    // of in front of an idle matrix pipe (2-3 us per tile), and the second tile of a CU needs no block launch.
    const int n_tiles_n = CoutP / PP_BN;
    int oy0, ox0, n0;
    int xoff[PP_XPW], woff[PP_FPW];
#define PP_SETUP(tile_)                                                                                \
    {                                                                                                  \
        const int swz_ = xcd_swizzle3((tile_), n_tiles);                                               \
        const int tn_ = swz_ % n_tiles_n, tsp_ = swz_ / n_tiles_n;                                     \
        const int tx_ = tsp_ % tiles_x, ty_ = tsp_ / tiles_x;                                          \
        oy0 = ty_ * PP_TH; ox0 = tx_ * PP_TW; n0 = tn_ * PP_BN;                                        \
        /* per-lane staging sources (element offsets; -1 = zero page) */                               \
        _Pragma("unroll") for (int i = 0; i < PP_XPW; ++i) {                                           \
            int piece = wave + 8 * i;                                                                  \
            if (piece >= PP_XCH) piece = PP_XCH - 1;                                                   \
            const int q = piece * 16 + (lane >> 2);                                                    \
            const int slot = (lane & 3) ^ ((q >> 2) & 3);                                              \
            int off = -1;                                                                              \
            if (q < PP_NPIX) {                                                                         \
                const int py = q / PP_PW, px = q - py * PP_PW;                                         \
                const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;                                        \
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) off = (iy * W + ix) * Cin + slot * 8;      \
            }                                                                                          \
            /* RING3: BYTE offsets for buffer_load ... lds; outside the image = beyond the descriptor's range, which reads zeros */ \
            xoff[i] = RING3 ? (off >= 0 ? off * 2 : (int)0x80000000) : off;                            \
        }                                                                                              \
        _Pragma("unroll") for (int i = 0; i < PP_FPW; ++i) {                                           \
            const int r = (wave * PP_FPW + i) * 16 + (lane >> 2);   /* row of the stage tile: tap r / 128, filter r % 128 */ \
            const int slot = (lane & 3) ^ ((r >> 2) & 3);                                              \
            /* KXM: the stage's three taps are (ky = r / 128, kx = stage % 3), three filter rows apart in the packed array */ \
            woff[i] = ((r / PP_BN) * (KXM ? 3 : 1) * CoutP + n0 + (r % PP_BN)) * PP_CC + slot * 8;     \
        }                                                                                              \
    }
    int tile = blockIdx.x;
    // RING3: the two input planes as buffer descriptors (patch copies by buffer_load ... lds, per-lane byte offsets)
    const int plane_bytes = (int)((size_t)H * W * Cin * sizeof(half_t));
    const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(in), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(RING3 ? in_c : in), 0, plane_bytes, 0x00020000);
    constexpr bool RANGE = (COMP & 2) && !(COMP & 4);      // compensated output: keep the largest value in front of the saturation
    unsigned int smax = 0;                                 // (wave-uniform: a scalar register across the tiles)
#ifdef SFD2_PP_CU_STAGGER
    // experiment builds: some CUs start late so that the tiles' epilogue write bursts of the short-K layer (conv2a) are not in phase
    // chip-wide.  SFD2_PP_CU_STAGGER = delay in 10 ns units, SFD2_PP_CU_STAGGER_WHO: 0 = the blocks with one tile less, 1 = every other
    // group of eight blocks.  Measured (tools/ab_libs.py, ms per extract, three rounds): as is 1.668 / 1.683 / 1.680; 19 us on the
    // blocks with one tile less 1.687 / 1.685 / 1.685; 19 us on every other group 1.672 / 1.679 / 1.688; 10 us on every other group
    // 1.719 / 1.725 / 1.718 -- phase diversity between CUs buys nothing here, not enabled
    if (Cin == 64) {
        const int rem_ = n_tiles % (int)gridDim.x;
        const bool late_ = SFD2_PP_CU_STAGGER_WHO ? ((blockIdx.x >> 3) & 1) : (rem_ != 0 && (int)blockIdx.x >= rem_);
        if (late_) { const unsigned long long t0_ = wall_clock64(); while (wall_clock64() - t0_ < SFD2_PP_CU_STAGGER) __builtin_amdgcn_s_sleep(32); }
    }
#endif
    PP_WALL(0)
    PP_SETUP(tile)
    const int NCH = Cin / PP_CC;                           // chunks per plane

#define PP_ISSUE_X1(chunk_, buf_, i_)                                                                  \
    do {                                                                                               \
        const int pc_ = (wave + 8 * (i_) < PP_XCH) ? wave + 8 * (i_) : PP_XCH - 1;                     \
        if constexpr (RING3) {   /* one 32-bit offset per piece and lane instead of a 64-bit address per piece, lane AND plane */ \
            const bool corr_ = ILV ? (((chunk_) & 1) != 0) : ((chunk_) >= NCH);                        \
            const int so_ = (ILV ? ((chunk_) >> 1) : (corr_ ? (chunk_) - NCH : (chunk_))) * (PP_CC * 2); \
            if (corr_)                                                                                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (lds_void3_t *)(Xs + (buf_)*PP_XBYTES + pc_ * 1024), 16, xoff[i_], so_, 0, 0); \
            else                                                                                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_h, (lds_void3_t *)(Xs + (buf_)*PP_XBYTES + pc_ * 1024), 16, xoff[i_], so_, 0, 0); \
            break;                                                                                     \
        }                                                                                              \
        /* X3: chunks [0, 2 NCH) read the hi plane (against the hi, then the lo' filters), [2 NCH, 3 NCH) the lo' plane */ \
        /* RING3: chunk 2 k = channels 32 k of the hi plane, 2 k + 1 = the same channels of the corr plane */ \
        const int pch_ = ILV ? ((chunk_) >> 1)                                                         \
                               : (COMP & 4) ? ((chunk_) >= NCH ? (chunk_) - NCH - ((chunk_) >= 2 * NCH ? NCH : 0) : (chunk_)) : (chunk_); \
        const half_t *pl_ = ILV ? (((chunk_) & 1) ? in_c : in)                                         \
                          : (COMP & 4) ? ((chunk_) >= 2 * NCH ? in_c : in)                             \
                                       : (((COMP & 1) && (chunk_) >= NCH) ? in_c - (size_t)NCH * PP_CC : in); \
        const half_t *src_ = xoff[i_] >= 0 ? pl_ + (size_t)xoff[i_] + pch_ * PP_CC : zero_page + (lane & 3) * 8; \
        __builtin_amdgcn_global_load_lds((gbl_void3_t *)src_,                                          \
                                         (lds_void3_t *)(Xs + (buf_)*PP_XBYTES + pc_ * 1024), 16, 0, 0); \
    } while (0)
#define PP_ISSUE_F1(stage_, buf_, i_)                                                                  \
    {                                                                                                  \
        /* X3: the filter array is [hi chunks][lo' chunks]; the K loop's chunk sequence uses hi, lo', hi */ \
        /* ILV: the array stays [hi chunks][corr chunks]; K-loop chunk c is array chunk (c & 1) * NCH + (c >> 1) */ \
        const int fst_ = ILV ? ((((stage_) / 3) & 1) * NCH + (((stage_) / 3) >> 1)) * 3 + (stage_) % 3   \
                             : (COMP & 4) ? ((stage_) >= 6 * NCH ? (stage_) - 6 * NCH : (stage_)) : (stage_); \
        const half_t *src_ = wpk + (size_t)(KXM ? (fst_ / 3) * 9 + fst_ % 3 : fst_ * 3) * CoutP * PP_CC + woff[i_]; \
        __builtin_amdgcn_global_load_lds((gbl_void3_t *)src_,                                          \
                                         (lds_void3_t *)(Fs + (buf_)*PP_FBYTES + (wave * PP_FPW + (i_)) * 1024), 16, 0, 0); \
    }
#define PP_ISSUE_F(stage_, buf_)                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < PP_FPW; ++i_) PP_ISSUE_F1(stage_, buf_, i_)

    const int NCT = ((COMP & 4) ? 3 : ((COMP & 1) ? 2 : 1)) * NCH;   // chunks of the K loop: the hi plane's, then the corr plane's (X3: three passes)
    const int NST = NCT * 3;

#pragma unroll
    for (int i = 0; i < PP_XPW; ++i) PP_ISSUE_X1(0, 0, i);
    PP_ISSUE_F(0, 0)
    if (RING3) { PP_ISSUE_F(1, 1) }
    for (int t = tid; t < PP_BN; t += 512) {
        SSb[t] = scale[n0 + t]; SSb[PP_BN + t] = shift[n0 + t];
        if (F6) SSb[2 * PP_BN + t] = shift[CoutP + n0 + t];      // (the scale bytes follow the shifts: launcher)
    }

    const int lrow = lane & 31, lhi = lane >> 5;
    int a_off[2], a_sw[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int r = wch + ct * 32 + lrow;
        a_off[ct] = r * 64;
        a_sw[ct] = (r >> 2) & 3;
    }
    const int qb = wrow * PP_PW + lrow;

    for (int it = 0;; ++it) {
    float *SS = SSb + (it & 1) * SSN * PP_BN;
    int sa6v[2] = {0, 0};                                  // F6: this lane's filter rows' scale bytes (published by the barriers of the K loop)
    f32x16_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    PP_CYC(0)
    SFD2_BARRIER_DRAIN();
    if (STAGGER && grp == 1) __builtin_amdgcn_s_barrier();
    PP_CYC(1)

    /* End of a stage (both sections of its last unit): what the NEXT stage reads must have landed -- its filters, and after a chunk's \
       last stage the next chunk's patch.  Two buffers: everything requested so far (the youngest request is a unit old).  RING3: the \
       requests younger than that may stay in flight -- per wave and chunk the queue is  X0 F F F X1 | X2 F F F X3 | X4 F F F  (F = the \
       wave's three pieces of stage st + 2), so behind F(st + 1) sit 5 / 6 / 3 requests at the end of stage 0 / 1 / 2 (stage 2 also needs \
       X4: only its own F stays out).  The last chunk requests nothing of the kind: full wait. */ \
#define PP_STAGE_WAIT(sg_, extra_)                                                                     \
    if (RING3 && more_x) {                                                                             \
        if ((sg_) == 0) asm volatile("s_waitcnt vmcnt(5) " extra_ ::: "memory");                       \
        else if ((sg_) == 1 && SPREAD) asm volatile("s_waitcnt vmcnt(5) " extra_ ::: "memory");        \
        else if ((sg_) == 1) asm volatile("s_waitcnt vmcnt(6) " extra_ ::: "memory");                  \
        else asm volatile("s_waitcnt vmcnt(3) " extra_ ::: "memory");                                  \
    } else asm volatile("s_waitcnt vmcnt(0) " extra_ ::: "memory");

#define PP_CHUNK_BODY(F8_)  /* one 32-channel chunk of the K loop; F8_: a corr-plane chunk (fp8 MFMA) */ \
        const unsigned char *xs = Xs + (c & 1) * PP_XBYTES; \
        const bool more_x = c + 1 < NCT; \
        constexpr bool f8 = (F8_) != 0; /* block-uniform */ \
        h8_t fr[2][6]; /* KXM: the stage's six patch rows */ \
        v8i_t frc[6]; \
_Pragma("unroll") \
        for (int t9 = 0; t9 < 9; ++t9) { \
/* u3 = unit within the stage (the stage's last unit carries the waits), sg = stage within the chunk */ \
            const int sg = t9 / 3, u3 = t9 % 3; \
            const int ky = KXM ? u3 : sg, kx = KXM ? sg : u3; \
            const int st = c * 3 + sg; \
            const unsigned char *fs = Fs + (RING3 ? sg : (st & 1)) * PP_FBYTES + u3 * (PP_BN * 64); /* RING3: st % 3 = sg */ \
/* ---------------- LOAD section */ \
            PP_UCYC(0) \
            h8_t fa[2][2], fb[2][4]; \
            if (ABL & 2) { /* timing ablation: no fragment reads */ \
_Pragma("unroll") \
                for (int kk = 0; kk < 2; ++kk) { \
_Pragma("unroll") \
                    for (int pr = 0; pr < 4; ++pr) fb[kk][pr] = h8_t{(half_t)(float)lane, 0, 0, 0, 0, 0, 0, 0}; \
_Pragma("unroll") \
                    for (int ct = 0; ct < 2; ++ct) fa[kk][ct] = h8_t{(half_t)(float)c, 0, 0, 0, 0, 0, 0, 0}; \
                } \
            } \
            int qv = qb; \
            asm volatile("" : "+v"(qv)); /* recompute the 8 patch addresses per unit (hoisted out of the loop they are 72 registers) */ \
            if (f8) { /* corr chunk: 8-dword pixel fragments (K slices 0 and 1 adjacent), the stage's six patch rows as in the fp16 loop */ \
_Pragma("unroll") \
                for (int j = (u3 == 0 ? 0 : 3 + u3); j < 4 + u3; ++j) { \
                    const int q = qv + j * PP_PW + kx; \
                    const int sw = (q >> 2) & 3; \
                    frc[j] = sfd2_cat8(*reinterpret_cast<const h8_t *>(xs + q * 64 + (((0 * 2 + lhi) ^ sw) << 4)), \
                                       *reinterpret_cast<const h8_t *>(xs + q * 64 + (((1 * 2 + lhi) ^ sw) << 4))); \
                } \
            } else if (KXM && !(ABL & 2)) { \
_Pragma("unroll") \
                for (int j = (u3 == 0 ? 0 : 3 + u3); j < 4 + u3; ++j) { /* rows 0..3, then 4, then 5 */ \
                    const int q = qv + j * PP_PW + kx; \
                    const int sw = (q >> 2) & 3; \
_Pragma("unroll") \
                    for (int kk = 0; kk < 2; ++kk) \
                        fr[kk][j] = *reinterpret_cast<const h8_t *>(xs + q * 64 + (((kk * 2 + lhi) ^ sw) << 4)); \
                } \
            } \
_Pragma("unroll") \
            for (int pr = 0; pr < ((ABL & 2) || KXM ? 0 : 4); ++pr) { \
                const int q = qv + (pr + ky) * PP_PW + kx; \
                const int sw = (q >> 2) & 3; \
_Pragma("unroll") \
                for (int kk = 0; kk < 2; ++kk) \
                    fb[kk][pr] = *reinterpret_cast<const h8_t *>(xs + q * 64 + (((kk * 2 + lhi) ^ sw) << 4)); \
            } \
            v8i_t fac[2]; \
            if (f8) { \
_Pragma("unroll") \
                for (int ct = 0; ct < 2; ++ct) \
                    fac[ct] = sfd2_cat8(*reinterpret_cast<const h8_t *>(fs + a_off[ct] + (((0 * 2 + lhi) ^ a_sw[ct]) << 4)), \
                                        *reinterpret_cast<const h8_t *>(fs + a_off[ct] + (((1 * 2 + lhi) ^ a_sw[ct]) << 4))); \
            } else { \
_Pragma("unroll") \
            for (int ct = 0; ct < ((ABL & 2) ? 0 : 2); ++ct) \
_Pragma("unroll") \
                for (int kk = 0; kk < 2; ++kk) \
                    fa[kk][ct] = *reinterpret_cast<const h8_t *>(fs + a_off[ct] + (((kk * 2 + lhi) ^ a_sw[ct]) << 4)); \
            } \
            if (!RING3 && !(ABL & 1) && u3 == 0 && st + 1 < NST) { PP_ISSUE_F(st + 1, (st + 1) & 1) } \
            if (!(ABL & 1) && more_x) { \
                if (t9 == 0) PP_ISSUE_X1(c + 1, (c + 1) & 1, 0); \
                if (t9 == 1) PP_ISSUE_X1(c + 1, (c + 1) & 1, 1); \
                if (t9 == 3) PP_ISSUE_X1(c + 1, (c + 1) & 1, 2); \
                if (t9 == 4) PP_ISSUE_X1(c + 1, (c + 1) & 1, 3); \
                if (t9 == 6) PP_ISSUE_X1(c + 1, (c + 1) & 1, 4); \
            } \
            /* RING3: stage st + 2 into the buffer stage st - 1 has just left, BEHIND this unit's patch piece in the wave's queue (the counted \
               waits below rely on the order  X0 F F F X1 | X2 F F F X3 | X4 F F F  per chunk) */ \
            if (RING3 && !SPREAD && u3 == 0 && st + 2 < NST) { PP_ISSUE_F(st + 2, (sg + 2) % 3) } \
            if (SPREAD && st + 2 < NST) { PP_ISSUE_F1(st + 2, (sg + 2) % 3, u3) }   /* queue per chunk: X0 F | X1 F | F || X2 F | X3 F | F || X4 F | F | F */ \
            if (u3 == 2) { PP_STAGE_WAIT(sg, "lgkmcnt(0)") } \
            PP_UCYC(1) \
            __builtin_amdgcn_sched_barrier(0); \
            if (!(PP_ABL_BARRIER && f8 && u3 == 1)) asm volatile("s_barrier" ::: "memory"); /* PP_ABL_BARRIER: timing experiment, wrong results */ \
            __builtin_amdgcn_sched_barrier(0); \
/* ---------------- MFMA section */ \
            PP_UCYC(2) \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
            if (PRIO) __builtin_amdgcn_s_setprio(1); \
            if (f8) { \
                static_assert(!(COMP & 1) || KXM, "the compensated instantiations use the column-major tap order"); \
_Pragma("unroll") \
                for (int ct = 0; ct < 2; ++ct) \
_Pragma("unroll") \
                    for (int pr = 0; pr < 4; ++pr) \
                        if (!PP_NO_CORR_MFMA) \
                        acc[ct][pr] = B6 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fac[ct], frc[pr + u3], acc[ct][pr], 2, SFD2_PIX6_BLGP, 0, sa6v[ct], 0, frc[pr + u3][6]) \
                                    : F6 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fac[ct], frc[pr + u3], acc[ct][pr], 2, 0, 0, sa6v[ct], 0, 0x7f7f7f7f) \
                                         : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fac[ct], frc[pr + u3], acc[ct][pr], 0, 0, 0, sa, 0, 0x7f7f7f7f); \
                /* the scaled MFMA is a pure node to instruction selection, which otherwise sinks all 72 of a chunk below its last \
                   barrier (every fragment of nine units live at once); an empty asm on the accumulators keeps each unit's in its section */ \
_Pragma("unroll") \
                for (int ct = 0; ct < 2; ++ct) \
_Pragma("unroll") \
                    for (int pr = 0; pr < 4; ++pr) asm volatile("" : "+v"(acc[ct][pr])); \
            } else { \
_Pragma("unroll") \
            for (int kk = 0; kk < 2; ++kk) \
_Pragma("unroll") \
                for (int ct = 0; ct < 2; ++ct) \
_Pragma("unroll") \
                    for (int pr = 0; pr < 4; ++pr) \
                        acc[ct][pr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kk][ct], KXM && !(ABL & 2) ? fr[kk][pr + u3] : fb[kk][pr], \
                                                                             acc[ct][pr], 0, 0, 0); \
            } \
            if (PRIO) __builtin_amdgcn_s_setprio(0); \
            __builtin_amdgcn_sched_barrier(0); \
            PP_UCYC(3) \
            if (u3 == 2) { PP_STAGE_WAIT(sg, "") } \
/* group 1's last MFMA section has nobody left to hand the pipe to */ \
            if (!(STAGGER && grp == 1 && t9 == 8 && c + 1 == NCT) && !(PP_ABL_BARRIER && f8 && u3 == 1)) asm volatile("s_barrier" ::: "memory"); \
            __builtin_amdgcn_sched_barrier(0); \
        } \
    /* end of PP_CHUNK_BODY */
    if constexpr (ILV) {
        if (F6) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) sa6v[ct] = __float_as_int(SS[2 * PP_BN + wch + ct * 32 + lrow]);
        }
        for (int cc = 0; cc < NCH; ++cc) {
            { const int c = 2 * cc; PP_CHUNK_BODY(0) }
            { const int c = 2 * cc + 1; PP_CHUNK_BODY(1) }
        }
    } else {
    for (int c = 0; c < NCH; ++c) { PP_CHUNK_BODY(0) }
    if (COMP & 1) {
        if (F6) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) sa6v[ct] = __float_as_int(SS[2 * PP_BN + wch + ct * 32 + lrow]);
        }
        for (int c = NCH; c < NCT; ++c) { PP_CHUNK_BODY(1) }
    }
    }
    if (COMP & 4) {
        // X3 (SFD2_PREC_F16X3 on pre-split planes): y = sum hi * w_hi + 2^-11 (sum hi * w_lo' + sum lo' * w_hi), the low parts
        // being stored scaled by 2^11 -- ONE accumulator: the first pass's sums are scaled up (exactly) before the cross terms
        // join them, 2^-11 goes into the epilogue
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] *= 2048.0f;
        for (int c = NCH; c < NCT; ++c) { PP_CHUNK_BODY(0) }
    }
#undef PP_CHUNK_BODY
#undef PP_STAGE_WAIT
    PP_CYC(2)
    // the next tile's first copies go out before this tile's epilogue (buffers 0: last read a chunk / a stage ago)
    const int eoy0 = oy0, eox0 = ox0, en0 = n0;
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < n_tiles;
    if (has_next) {
        PP_SETUP(next)
#pragma unroll
        for (int i = 0; i < PP_XPW; ++i) PP_ISSUE_X1(0, 0, i);
        PP_ISSUE_F(0, 0)
        if (RING3) { PP_ISSUE_F(1, 1) }     // (buffers 0 and 1: the last stage sat in buffer 2, stage NST - 2's readers are barriers behind)
        float *SSn = SSb + ((it + 1) & 1) * SSN * PP_BN;
        for (int t = tid; t < PP_BN; t += 512) {
            SSn[t] = scale[n0 + t]; SSn[PP_BN + t] = shift[n0 + t];
            if (F6) SSn[2 * PP_BN + t] = shift[CoutP + n0 + t];
        }
    }

    const float lo = relu ? 0.0f : -__builtin_huge_valf();   // ReLU without a branch (this file is compiled with -fno-honor-nans:
                                                              // no canonicalising v_max in front of each fmaxf)
    // epilogue: y = acc * scale + shift (ReLU), regrouped with v_permlane32_swap into 16-byte stores
    // (see conv2_kernels.hip)
    float mx = 0.0f;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int oy = eoy0 + wrow + pr, ox = eox0 + lrow;
        const bool inb = oy < Ho && ox < Wo;
        const size_t pix = S2D ? ((size_t)((inb ? oy : 0) >> 1) * (Wo >> 1) + ((inb ? ox : 0) >> 1)) * 4 + (((inb ? oy : 0) & 1) * 2 + ((inb ? ox : 0) & 1))
                               : (size_t)(inb ? oy : 0) * Wo + (inb ? ox : 0);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int cl = wch + ct * 32 + 4 * lhi;
            if constexpr (O6) {
                // fp6 corr records (sfd2_epi16_fp6): this lane's 16 channels of the (pixel, 32-channel chunk) are one half-record
                float4 sc4[4], sh4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    sc4[q] = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                    sh4[q] = *reinterpret_cast<const float4 *>(SS + PP_BN + cl + 8 * q);
                }
                uint2 hv4[4];
                uint4 r0, r1;
                sfd2_epi16_fp6(acc[ct][pr], sc4, sh4, (relu & 1) ? 0.0f : -SFD2_C_SAT, hv4, r0, r1, mx, inb);
                const size_t ob = pix * CoutP + en0 + wch + ct * 32;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const auto t0 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].x, hv4[2 * m + 1].x, false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].y, hv4[2 * m + 1].y, false, false);
                    if (inb) *reinterpret_cast<uint4 *>(out + ob + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
                if (inb) {
                    *reinterpret_cast<uint4 *>(out_c + ob + 8 * lhi) = r0;            // slot lhi
                    *reinterpret_cast<uint4 *>(out_c + ob + 16 + 8 * lhi) = r1;       // slot 2 + lhi
                }
                continue;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const size_t o16 = pix * CoutP + en0 + wch + ct * 32 + 8 * (2 * m + lhi);
                uint2 pk[2], ck[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const float4 sc = *reinterpret_cast<const float4 *>(SS + cl + 8 * q);
                    const float4 sh = *reinterpret_cast<const float4 *>(SS + PP_BN + cl + 8 * q);
                    if (COMP & 4) {
                        constexpr float k = 1.0f / 2048.0f;
                        float v0 = acc[ct][pr][4 * q + 0] * k * sc.x + sh.x, v1 = acc[ct][pr][4 * q + 1] * k * sc.y + sh.y;
                        float v2 = acc[ct][pr][4 * q + 2] * k * sc.z + sh.z, v3 = acc[ct][pr][4 * q + 3] * k * sc.w + sh.w;
                        v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
                        if (COMP & 8) {          // fp32 output: this lane's four channels are 16 contiguous bytes
                            if (inb) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + pix * CoutP + en0 + cl + 8 * q) = make_float4(v0, v1, v2, v3);
                        } else {                 // hi / lo' planes (x3_split's arithmetic)
                            const h4_t hv = cvt4c(v0, v1, v2, v3);
                            const h4_t lv = cvt4c((v0 - (float)hv[0]) * 2048.0f, (v1 - (float)hv[1]) * 2048.0f, (v2 - (float)hv[2]) * 2048.0f,
                                                  (v3 - (float)hv[3]) * 2048.0f);
                            __builtin_memcpy(&pk[j], &hv, 8);
                            __builtin_memcpy(&ck[j], &lv, 8);
                        }
                    } else if (OR1) {
                        sfd2_epi4_r1<false>(acc[ct][pr][4 * q + 0], acc[ct][pr][4 * q + 1], acc[ct][pr][4 * q + 2], acc[ct][pr][4 * q + 3], sc, sh,
                                            sc, relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j].x, mx, inb);
                    } else if (COMP & 2) {
                        sfd2_epi4<false>(acc[ct][pr][4 * q + 0], acc[ct][pr][4 * q + 1], acc[ct][pr][4 * q + 2], acc[ct][pr][4 * q + 3], sc, sh,
                                         sc, relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j], mx, inb);
                    } else {
                        float v0 = acc[ct][pr][4 * q + 0] * sc.x + sh.x;
                        float v1 = acc[ct][pr][4 * q + 1] * sc.y + sh.y;
                        float v2 = acc[ct][pr][4 * q + 2] * sc.z + sh.z;
                        float v3 = acc[ct][pr][4 * q + 3] * sc.w + sh.w;
                        v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
                        const h4_t hv = cvt4c(v0, v1, v2, v3);
                        __builtin_memcpy(&pk[j], &hv, 8);
                    }
                }
                if ((COMP & 12) == 12) continue;   // (fp32 output: stored above)
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                if (inb && (!(ABL & 4) || t0[0] == 0x12345678u)) *reinterpret_cast<uint4 *>(out + o16) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                if (OR1) {          // eight residual bytes: this lane's channels 8 (2 m + lhi) .. + 7
                    const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                    if (inb) *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(out_c) + o16) = make_uint2(u0[0], u0[1]);
                } else if (COMP & 2) {
                    const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                    const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                    if (inb) *reinterpret_cast<uint4 *>(out_c + o16) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                }
            }
        }
    }
    if (RANGE) { const unsigned int wb = sfd2_wave_max_bits(mx); smax = wb > smax ? wb : smax; }
    PP_CYC(3)
    if (!has_next) break;
    tile = next;
    }
    if (RANGE) sfd2_range_commit(range, smax);
#ifdef SFD2_PP_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PP_WALL(1)
#endif
#undef PP_ISSUE_X1
#undef PP_ISSUE_F
#undef PP_SETUP
}

template <int STAGGER, int PRIO, int ABL = 0, int COMP = 0>
static void launch_pp_t(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int CoutP, int relu, half_t *out,
                        int Ho, int Wo, const half_t *zero_page, const half_t *in_c = nullptr, half_t *out_c = nullptr, int sa = 0,
                        unsigned int *range = nullptr)
{
    constexpr bool ring3 = (SFD2_PP_RING3 != 0) && (COMP & 1) && !(COMP & 4) && SFD2_PP_KXM && !ABL;
    constexpr size_t lds = (size_t)2 * PP_XBYTES + (size_t)(ring3 ? 3 : 2) * PP_FBYTES + ((COMP & 16) ? 6 : 4) * PP_BN * sizeof(float);
    static bool attr_done = false;
    auto kern = conv3x3_pp_kernel<STAGGER, PRIO, ABL, SFD2_PP_KXM, COMP>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        slots = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
                    ? cus : 256;                           // one resident block per CU
    }
    const int tiles_x = (Wo + PP_TW - 1) / PP_TW, tiles_y = (Ho + PP_TH - 1) / PP_TH;
    const int n_tiles = tiles_x * tiles_y * (CoutP / PP_BN);
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out,
                       Ho, Wo, tiles_x, n_tiles, zero_page, in_c, out_c, sa, range);
#ifdef SFD2_PP_TRACE
    {
        static int dumps = 0;
        if (H >= 250 && ++dumps == 40) {
            (void)hipStreamSynchronize(st);
            static unsigned long long hw[1024][2], hc[2][8][4];
            (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_pp_wall), sizeof(hw));
            (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_pp_cyc), sizeof(hc));
            static unsigned long long hu[2][2][9][4];
            (void)hipMemcpyFromSymbol(hu, HIP_SYMBOL(g_pp_unit), sizeof(hu));
            for (int w = 0; w < 2; ++w)
                for (int k = 0; k < 2; ++k) {
                    fprintf(stderr, "  wave %d %s chunk, per unit [LOAD, barrier wait, MFMA section, to next LOAD]:", w * 4, k ? "corr" : "hi  ");
                    for (int u = 0; u < 9; ++u)
                        fprintf(stderr, " [%lld %lld %lld %lld]", (long long)(hu[w][k][u][1] - hu[w][k][u][0]), (long long)(hu[w][k][u][2] - hu[w][k][u][1]),
                                (long long)(hu[w][k][u][3] - hu[w][k][u][2]), u < 8 ? (long long)(hu[w][k][u + 1][0] - hu[w][k][u][3]) : 0ll);
                    fprintf(stderr, "\n");
                }
            unsigned long long t0 = ~0ull, e0 = ~0ull, e1 = 0; double es = 0;
            for (int b = 0; b < grid; ++b) t0 = hw[b][0] < t0 ? hw[b][0] : t0;
            for (int b = 0; b < grid; ++b) { const unsigned long long e = hw[b][1] - t0; e0 = e < e0 ? e : e0; e1 = e > e1 ? e : e1; es += (double)e; }
            fprintf(stderr, "pp trace COMP %d  %dx%d Cin %d CoutP %d: %d tiles on %d blocks; exit (10 ns) min %llu mean %.0f max %llu\n", COMP, H, W, Cin, CoutP, n_tiles, grid, e0, es / grid, e1);
            for (int w = 0; w < 2; ++w)
                for (int t = 0; t < (n_tiles + grid - 1) / grid && t < 8; ++t)
                    fprintf(stderr, "  wave %d tile %d: wait %lld  K loop %lld  next-tile issue %lld  epilogue %lld   (cycles)\n", w * 4, t,
                            (long long)(hc[w][t][1] - hc[w][t][0]), (long long)(hc[w][t][2] - hc[w][t][1]), 0ll, (long long)(hc[w][t][3] - hc[w][t][2]));
        }
    }
#endif
}

// compensated instantiations (SFD2_PREC_F16C): wpk = the layer's wc array, sbyte its scale byte
void launch_conv3x3_pp_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                         const float *scale, const float *shift, int CoutP, int relu, half_t *out, half_t *out_c,
                         int Ho, int Wo, const half_t *zero_page, int sbyte, const float *shift_sa6, unsigned int *range, int fmt6)
{
    const int sa = (sbyte & 255) * 0x01010101;
    // option "c3b_plain": no corr plane on one side.  Plain input (conv3b): the K loop ends with the hi chunks, the epilogue still writes the output's corr
    // bytes (fmt6 bit 3: the three-byte trunk form).  Plain output (conv3a, its corr plane having no reader): fp6 x fp6 correction chunks, fp16 epilogue.
    if (!in_c && out_c) {
        if (fmt6 & 8) launch_pp_t<1, 1, 0, 2 | 256>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        else launch_pp_t<1, 1, 0, 2>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        return;
    }
    if (in_c && !out_c && shift_sa6 && (fmt6 & 1)) {
        launch_pp_t<1, 1, 0, 1 | 16 | 32>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        return;
    }
    // shift_sa6 != null: wpk's corr rows are fp6 strings and shift_sa6 = [shift[CoutP] | the rows' scale bytes as ints [CoutP]]
    if (in_c && out_c && shift_sa6 && fmt6 != 0) {      // fp6 corr records: bit 0 of fmt6 = the input's, bit 1 = the output's (filters: the (w, lo'_w) strings)
        if ((fmt6 & 3) == 3) launch_pp_t<1, 1, 0, 3 | 16 | 32 | 64>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
#ifdef SFD2_EXPERIMENTS
        else if ((fmt6 & 1) && sfd2_env("SFD2_PPC_ABL"))    // timing ablation (wrong results): conv3b's instantiation without its staging copies
            launch_pp_t<1, 1, 1, 3 | 16 | 32>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
#endif
#ifdef SFD2_EXPERIMENTS
        else if ((fmt6 & 1) && sfd2_env("SFD2_PPC_ABL"))    // timing ablation (wrong results): conv3b's instantiation without its staging copies
            launch_pp_t<1, 1, 1, 3 | 16 | 32>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
#endif
        else if ((fmt6 & 9) == 9) launch_pp_t<1, 1, 0, 3 | 16 | 32 | 256>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        else if ((fmt6 & 5) == 5) launch_pp_t<1, 1, 0, 3 | 16 | 32 | 128>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        else if (fmt6 & 1) launch_pp_t<1, 1, 0, 3 | 16 | 32>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        else launch_pp_t<1, 1, 0, 3 | 16 | 64>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        return;
    }
    if (in_c && out_c && (fmt6 & 2)) { launch_pp_t<1, 1, 0, 3 | 64>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range); return; }
    if (in_c && out_c && shift_sa6) launch_pp_t<1, 1, 0, 19>(st, in, H, W, Cin, wpk, scale, shift_sa6, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
    else if (in_c && out_c) launch_pp_t<1, 1, 0, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
    else if (in_c) launch_pp_t<1, 1, 0, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa);
}

// SFD2_PREC_F16X3 on pre-split planes (hi = fp16(x), lo' = fp16((x - hi) * 2^11), x3_split's arithmetic): in / in_lo = the input's
// planes, wpk = the filters as [hi chunks][lo' chunks] in this kernel's packed layout; the output is either planes again
// (out_lo != null) or fp32 [Ho][Wo][CoutP] (out_f32).  Three passes of the fp16 K loop: hi x hi, hi x lo', lo' x hi.
void launch_conv3x3_pp_x3(hipStream_t st, const half_t *in, const half_t *in_lo, int H, int W, int Cin, const half_t *wpk,
                          const float *scale, const float *shift, int CoutP, int relu, half_t *out_hi, half_t *out_lo, float *out_f32,
                          int Ho, int Wo, const half_t *zero_page, int s2d)
{
    if (s2d && !out_f32) { launch_pp_t<1, 1, 0, 6 | 128>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out_hi, Ho, Wo, zero_page, in_lo, out_lo, 0); return; }
    if (out_f32) launch_pp_t<1, 1, 0, 12>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, reinterpret_cast<half_t *>(out_f32), Ho, Wo, zero_page, in_lo, nullptr, 0);
    else launch_pp_t<1, 1, 0, 6>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out_hi, Ho, Wo, zero_page, in_lo, out_lo, 0);
}

// does conv3x3_pp serve this layer?  (decided from the layer's shape alone: the filters are packed for it)
// Measured at 1600x1200 (tools/ab_libs.py, interleaved A/B on one box): conv3b 132.7 -> 122.3 us, convDa.0 134.5 -> 126.1,
// convDa.3 132.7 -> 122.2, conv3a 77.3 -> 72.8, convPa.3 60.4 -> 54.5; the same tile WITHOUT the one-barrier offset: 144 us
// (experiment builds: SFD2_CONV_PP_NOSTAGGER=1).  conv2a (64 -> 128 channels, two K chunks) is slower here (91 vs 81 us: the
// 63 KB prologue is a fifth of its K loop; with the persistent blocks 87.5 vs 82.7) and stays on conv_igemm2.
bool conv3x3_pp_serves(int ks, int stride, int CoutP, int Cin)
{
    static const bool off = sfd2_env("SFD2_CONV_NO_PP") != nullptr;   // experiment builds: conv_igemm2 for these layers
    static const bool c128 = sfd2_env("SFD2_PP_COUT128") != nullptr;   // experiment: conv2a (64 -> 128 channels) too
    return !off && ks == 3 && stride == 1 && (CoutP % 256 == 0 || (c128 && CoutP % 128 == 0)) && Cin % 64 == 0;
}

void launch_conv3x3_pp(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                       const float *scale, const float *shift, int CoutP, int relu, half_t *out,
                       int Ho, int Wo, const half_t *zero_page)
{
#ifdef SFD2_EXPERIMENTS
    if (const char *ab = sfd2_env("SFD2_PP_ABL")) {   // timing ablations (wrong results): 1 no copies, 2 no reads, 4 no stores
        switch (atoi(ab)) {
        case 4: launch_pp_t<1, 1, 4>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page); return;
        case 7: launch_pp_t<1, 1, 7>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page); return;
        case 3: launch_pp_t<1, 1, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page); return;
        default: break;
        }
    }
    static const bool nostagger = sfd2_env("SFD2_CONV_PP_NOSTAGGER") != nullptr;
    if (nostagger) {
        launch_pp_t<0, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
        return;
    }
#endif
    launch_pp_t<1, 1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
}
