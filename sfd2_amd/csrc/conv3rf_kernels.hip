// conv3x3_rf: implicit-GEMM kernel for the 3x3 layers with >= 256 output channels whose output is SMALL (stride-2
// convPa.0, and convPa.3 behind it: 150 x 200 pixels at 1600x1200; every 256-channel 3x3 layer of a 640x480 image).
// nets/sfd2.py:285-297.  Same GEMM orientation, record swizzle, filter packing and epilogue as conv3_kernels.hip.
//
// With 30 000 output pixels the 512-pixel x 128-channel tiles of conv3x3_pp give 140 blocks for 256 CUs (and the
// stride-2 kernel's 4-row tiles 266: two rounds, the second one almost empty).  This kernel's tile is 128 pixels
// (8 rows x 16) x ALL 256 channels: 247 blocks for 150 x 200.  A tile that small cannot afford to stage the filters in
// LDS (147 KB per K chunk for 128 pixels: the ~18 B/clk LDS-DMA path would take twice as long as the chunk's MFMAs),
// so the FILTERS NEVER TOUCH THE LDS: each of the eight waves owns 32 output channels and loads its A fragments straight
// from the packed filter array (L2-resident, 1.2 MB per layer) into registers, one whole chunk (nine units of tap x 32
// channels) ahead in a ring of nine: vector-memory operations complete in issue order, so a counted wait for a filter
// load also waits for every patch copy issued before it -- the ring gives those copies nine units to land.  Only the
// input patch goes through the LDS (buffer_load ... lds, three buffers of one 32-channel chunk each, the patch two
// chunks ahead requested one piece per unit); every wave reads all four pixel fragments of it.
//
//   per unit and wave: 2 global_load_dwordx4 (A), 8 ds_read_b128 (B), 8 MFMAs 32x32x16; B lives in a ring of three K
//   halves (the reads of half h + 2 are issued in front of the MFMAs of half h); one workgroup barrier per chunk, in
//   front of the chunk's last unit.
//
// Patch records (64 B = 32 channels of one pixel, 16-byte slots XOR-swizzled by (record >> 2) & 3 as everywhere):
//   stride 1: record = row * 32 + col                     (10 rows x 18 columns used)
//   stride 2: record = row * 40 + (col even ? col / 2 : 17 + col / 2)   (17 rows x 33 columns, de-interleaved by column
//             parity so that the 16 pixels of a fragment row are 16 CONSECUTIVE records for every tap)
// A pixel fragment is 2 output rows x 16 pixels; the row pitches (32, 40) make the two half-fragments 0 mod 16 records
// apart, which is what keeps the ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS) conflict-free.
#include "sfd2_internal.h"
#include <type_traits>

#define RF_TH 8
#define RF_TW 16
#define RF_BN 256
#define RF_CC 32

typedef __attribute__((address_space(3))) void lds_void4_t;
typedef const __attribute__((address_space(1))) void gbl_void4_t;

template <int S>
struct RfGeom {
    static constexpr int PH = (RF_TH - 1) * S + 3;        // 10 / 17 patch rows
    static constexpr int PWU = (RF_TW - 1) * S + 3;       // 18 / 33 used records per row
    static constexpr int P = S == 2 ? 40 : 32;            // row pitch in records
    static constexpr int NREC = PH * P;
    static constexpr int NPIECE = (NREC + 15) / 16;       // 20 / 43 pieces of 1 KB
    static constexpr int PPW = (NPIECE + 7) / 8;          // pieces per wave (the tail repeats the last piece)
    static constexpr int XBYTES = NPIECE * 1024;
    static constexpr int DF = 2 * S * P;                  // records between two fragments (two output rows)
    // vector-memory operations a wave issues after the last piece of a patch and before the chunk-boundary wait for it
    // one chunk later: the filter loads of units PPW-1..8 and 0..7 (2 each) and the PPW pieces of the patch after it
    static constexpr int VMW = 2 * (10 - PPW) + 16 + PPW;
};

__device__ __forceinline__ int xcd_swizzle4(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

__device__ __forceinline__ h4_t cvt4r(float a, float b, float c, float d)
{
    h4_t r;
    r[0] = (half_t)a; r[1] = (half_t)b; r[2] = (half_t)c; r[3] = (half_t)d;
    return r;
}

// one LDS-DMA piece: 64 lanes x 16 bytes from base + voff (+ soff) to 1 KB of LDS at dst; offsets beyond nbytes read zeros.
// (buffer_load ... lds, not global_load ... lds: hipcc books the latter as a FLAT access that may complete out of order
// and from then on turns every counted wait for a filter load into vmcnt(0) -- one full drain per chunk.)
__device__ __forceinline__ void rf_copy_piece(const half_t *base, int nbytes, unsigned char *dst, int voff, int soff)
{
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(base), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void4_t *)dst, 16, voff, soff, 0, 0);
}

// ABL: timing ablations for experiment builds (wrong results): 1 = no filter loads in the loop, 2 = no fragment reads,
// 4 = no patch copies in the loop, 8 = no output stores
//
// BN = output channels per block = the layer's padded channel count: 256 (eight waves x 32 channels, every wave all four
// pixel fragments) or 128 (four channel groups x two pixel halves of two fragments: conv2b).
//
// PERSISTENT blocks (one per CU): tiles blockIdx.x, + gridDim.x, ...  The chunk pipeline does not stop at a tile boundary:
// the patch ring (chunk C + 2 requested during chunk C), the filter ring (unit U + 9 loaded after unit U; the filters repeat
// from tile to tile) and the fragment prefetch run across it, and a tile's epilogue sits between the last unit of its last
// chunk and the first unit of the next tile's first chunk, whose operands are already in registers / in flight.  A
// one-tile-per-CU launch (convPa.0 / convPa.3 at 1600x1200) behaves as before; conv2b (938 tiles of four chunks) no longer
// pays a 230 KB prologue burst and an epilogue per 4.6k cycles of MFMA work with nothing running beside them.
//
// RES (conv2a: 64 -> 128 channels, two chunks = 18 units): the filter ring covers the WHOLE filter set (18 x 2 fragments =
// 144 registers per wave), so the filters are loaded once per block and never again -- the 128-pixel tile that cannot
// afford to stream 295 KB of filters per tile does not have to.  The chunk loop is unrolled over the two chunks so that the
// fragment index is a compile-time constant.
// COMP (SFD2_PREC_F16C, sfd2_internal.h): bit 0 = the input has a corr plane (in_c) and wpk holds 2 * Cin / 32 chunks: a tile's
// chunk sequence simply continues through the corr plane's chunks, whose units issue ONE v_mfma_scale_f32_32x32x64_f8f6f4 per
// pixel fragment instead of two fp16 MFMAs; bit 1 = the output's corr plane is written.  In these instantiations the filter
// ring and the pixel-fragment ring hold 8-dword tuples (both K slices of a unit adjacent: the fp8 MFMA's operand; the fp16
// MFMAs take the two halves), the pixel ring is two units deep and is prefetched one unit ahead.
// (A second accumulator set for conv2b's two-fragment waves -- four independent MFMA chains per wave instead of two, 254
// registers -- measured 141.2 against 141.0 us: the layer is bound by its filter stream, not by the chains; not kept.)
template <int S, int BN = RF_BN, int ABL = 0, bool RES = false, int COMP = 0>
__global__ __launch_bounds__(512, 2)
void conv3x3_rf_kernel(const half_t *__restrict__ in, int H, int W, int Cin,
                       const half_t *__restrict__ wpk, const float *__restrict__ scale,
                       const float *__restrict__ shift, int CoutP, int relu,
                       half_t *__restrict__ out, int Ho, int Wo, int tiles_x, int n_tiles,
                       const half_t *__restrict__ zero_page,
                       const half_t *__restrict__ in_c = nullptr, half_t *__restrict__ out_c = nullptr, int sa = 0,
                       unsigned int *__restrict__ range = nullptr /* the output tensor's range-status slot (compensated output) */)
{
    static_assert(COMP == 0 || !RES, "compensated instantiations: streamed filters");
    // (ABL with COMP = 3: experiment builds only, SFD2_RFC_ABL -- 1 no filter loads, 2 no fragment reads, 4 no patch copies, 8 no stores, 16 no MFMAs)
    // COMP bit 2 (X3, SFD2_PREC_F16X3): in / in_c are hi / lo' planes, wpk holds the filters' hi units then their lo' units; a tile's
    // chunk sequence runs the plain fp16 body three times -- (hi plane, hi filters), the accumulators times 2^11 (exact), (hi plane, lo'
    // filters), (lo' plane, hi filters) -- and the epilogue folds 2^-11 into the scale.  Output: hi / lo' planes (out / out_c), or with
    // bit 3 fp32 [P][CoutP] through `out`.  CPB = the compensated-structure bits (0 for X3: it is the plain kernel's pipeline).
    constexpr bool X3 = (COMP & 4) != 0;
    constexpr int CPB = X3 ? 0 : (COMP & 3);               // (bit 6, with CPB: the output's corr records as fp6 half-records)
    constexpr bool OUTC = (CPB & 2) != 0 || (X3 && !(COMP & 8));
    using G = RfGeom<S>;
    constexpr int NWC = BN / 32, NF = 4 / (8 / NWC);       // channel groups; pixel fragments per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Xs = smem;                              // [3][XBYTES]
    // (an epilogue staged through LDS -- whole 256-byte pixel rows, 1 KB contiguous per store instruction instead of 32 B of
    // 32 pixels -- was built for conv2a's 123 MB of output and measured: 77.7 vs 77.3 us, i.e. the stores are not what this
    // layer waits for; kept as an option)
    constexpr bool STG = false;
    unsigned char *Stage = smem + 3 * G::XBYTES;           // [128][BN] fp16 when STG
    (void)Stage;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lhi = lane >> 5, l32 = lane & 31, ly = l32 >> 4, lx = l32 & 15;
    const int f0 = (wave / NWC) * NF;
    const int n0 = (wave % NWC) * 32;                      // CoutP == BN: one channel tile (launcher)

    const int NCP = Cin / RF_CC;                           // chunks per plane
    const int NCH = ((CPB & 1) ? 2 : X3 ? 3 : 1) * NCP;    // chunks per tile: the hi plane's, then (COMP & 1) the corr plane's / (X3) two more passes
    const int NU1 = NCP * 9;                               // units of one pass
    const int NU = NCH * 9;
    const int n_my = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // tiles of this block
    const int TC = n_my * NCH;                             // chunks of this block

    // ---- per-lane staging sources of ONE tile: byte offsets into the input for buffer_load ... lds; padding and
    // out-of-image records get an offset beyond the buffer's range, for which the hardware returns zeros
    const int in_bytes = (int)((size_t)H * W * Cin * sizeof(half_t));
    int xoff[G::PPW];
    int xoff_seq = -1;                                     // which of this block's tiles xoff describes
#define RF_SETUP_X(seq_)                                                                               \
    {                                                                                                  \
        const int swz_ = xcd_swizzle4((int)blockIdx.x + (seq_) * (int)gridDim.x, n_tiles);             \
        const int tx_ = swz_ % tiles_x, ty_ = swz_ / tiles_x;                                          \
        const int poy0_ = ty_ * RF_TH, pox0_ = tx_ * RF_TW;                                            \
        int lane_ = lane;                                                                              \
        if (CPB != 0) asm volatile("" : "+v"(lane_));   /* (per-lane piece geometry recomputed per tile, not hoisted into registers) */ \
        _Pragma("unroll") for (int i = 0; i < G::PPW; ++i) {                                           \
            int piece = wave + 8 * i;                                                                  \
            if (piece >= G::NPIECE) piece = G::NPIECE - 1;                                             \
            const int q = piece * 16 + (lane_ >> 2);                                                   \
            const int slot = (lane_ & 3) ^ ((q >> 2) & 3);                                             \
            const int row = q / G::P, ir = q - row * G::P;                                             \
            int off = (int)0x80000000;                                                                 \
            if (row < G::PH && ir < G::PWU) {                                                          \
                const int col = S == 2 ? (ir < RF_TW + 1 ? 2 * ir : 2 * (ir - (RF_TW + 1)) + 1) : ir;  \
                const int iy = poy0_ * S - 1 + row, ix = pox0_ * S - 1 + col;                          \
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) off = ((iy * W + ix) * Cin + slot * 8) * (int)sizeof(half_t); \
            }                                                                                          \
            xoff[i] = off;                                                                             \
        }                                                                                              \
        xoff_seq = (seq_);                                                                             \
    }

#define RF_ISSUE_X1(chunk_, buf_, i_)                                                                  \
    do {                                                                                               \
        const int pc_ = (wave + 8 * (i_) < G::NPIECE) ? wave + 8 * (i_) : G::NPIECE - 1;               \
        const bool cp_ = X3 ? (chunk_) >= 2 * NCP : ((CPB & 1) && (chunk_) >= NCP);                    \
        const int ck_ = X3 ? ((chunk_) >= 2 * NCP ? (chunk_) - 2 * NCP : ((chunk_) >= NCP ? (chunk_) - NCP : (chunk_))) \
                           : (chunk_) - (cp_ ? NCP : 0);                                               \
        rf_copy_piece(cp_ ? in_c : in, in_bytes, Xs + (buf_)*G::XBYTES + pc_ * 1024, xoff[i_],         \
                      ck_ * RF_CC * (int)sizeof(half_t));                                              \
    } while (0)
#define RF_ISSUE_X(chunk_, buf_)                                                                       \
    _Pragma("unroll") for (int i_ = 0; i_ < G::PPW; ++i_) RF_ISSUE_X1(chunk_, buf_, i_);

    // A fragments: lane -> filter row n0 + (lane & 31), K slice (lane >> 5) + 2 * kk of the unit's [CoutP][32] tile
    int aoff = (n0 + l32) * RF_CC + lhi * 8;
#define RF_LOAD_A(u_, dst_)                                                                            \
    do {                                                                                               \
        int ao_ = aoff;                                                                                \
        asm volatile("" : "+v"(ao_));                                                                  \
        const int uu_ = (X3 && (u_) >= 2 * NU1) ? (u_) - 2 * NU1 : (u_);   /* third pass: the hi filters again */ \
        const half_t *ap_ = wpk + (size_t)uu_ * CoutP * RF_CC + ao_;                                   \
        dst_[0] = *reinterpret_cast<const h8_t *>(ap_);                                                \
        dst_[1] = *reinterpret_cast<const h8_t *>(ap_ + 16);                                           \
    } while (0)

    // B fragments: record of (fragment 0, tap) for this lane; fragment f is f * DF records further
    const int qb = ly * S * G::P + lx;
#define RF_READ_B(xs_, t_, kk_, dst_)                                                                  \
    do {                                                                                               \
        const int ky_ = (t_) / 3, kx_ = (t_) % 3;                                                      \
        const int to_ = ky_ * G::P + (S == 2 ? (kx_ == 0 ? 0 : (kx_ == 1 ? RF_TW + 1 : 1)) : kx_);     \
        int q_ = qb;                                                                                   \
        asm volatile("" : "+v"(q_));                                                                   \
        q_ += to_;                                                                                     \
        const unsigned char *bp_ = (xs_) + q_ * 64 + (((((kk_)*2 + lhi)) ^ ((q_ >> 2) & 3)) << 4);    \
        _Pragma("unroll") for (int f_ = 0; f_ < NF; ++f_)                                              \
            dst_[f_] = *reinterpret_cast<const h8_t *>(bp_ + (f0 + f_) * (G::DF * 64));                \
    } while (0)

    // the same as 8-dword tuples (COMP): K slices 0 and 1 of a fragment in adjacent registers
#define RF_LOAD_A8(u_, dst_)                                                                           \
    do {                                                                                               \
        int ao_ = aoff;                                                                                \
        asm volatile("" : "+v"(ao_));                                                                  \
        const half_t *ap_ = wpk + (size_t)(u_)*CoutP * RF_CC + ao_;                                    \
        dst_ = sfd2_cat8(*reinterpret_cast<const h8_t *>(ap_), *reinterpret_cast<const h8_t *>(ap_ + 16)); \
    } while (0)
#define RF_READ_B8(xs_, t_, dst_)                                                                      \
    do {                                                                                               \
        const int ky_ = (t_) / 3, kx_ = (t_) % 3;                                                      \
        const int to_ = ky_ * G::P + (S == 2 ? (kx_ == 0 ? 0 : (kx_ == 1 ? RF_TW + 1 : 1)) : kx_);     \
        int q_ = qb;                                                                                   \
        asm volatile("" : "+v"(q_));                                                                   \
        q_ += to_;                                                                                     \
        const unsigned char *b0_ = (xs_) + q_ * 64 + (((lhi) ^ ((q_ >> 2) & 3)) << 4);                 \
        const unsigned char *b1_ = (xs_) + q_ * 64 + (((2 + lhi) ^ ((q_ >> 2) & 3)) << 4);             \
        _Pragma("unroll") for (int f_ = 0; f_ < NF; ++f_)                                              \
            dst_[f_] = sfd2_cat8(*reinterpret_cast<const h8_t *>(b0_ + (f0 + f_) * (G::DF * 64)),      \
                                 *reinterpret_cast<const h8_t *>(b1_ + (f0 + f_) * (G::DF * 64)));     \
    } while (0)

    f32x16_t acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
    constexpr int NFA = RES ? 18 : 9;
    h8_t fa[NFA][2], fb[3][CPB ? 1 : NF];                // (COMP: the filter ring stays in K-slice halves -- 4-register pieces
    v8i_t fb8[2][CPB ? NF : 1];                          //  allocate where 8-register tuples spill -- and is joined per fp8 unit)

    // scale / shift of this wave's 32 channels stay in registers (one channel tile: the same for every tile); RES has no
    // registers to spare and re-reads them per tile
    float4 sc[4], sh[4];
    constexpr bool SS_RELOAD = RES || CPB != 0;   // (the compensated forms neither: two chunk bodies)
    if (!SS_RELOAD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = *reinterpret_cast<const float4 *>(scale + n0 + 4 * lhi + 8 * q);
            sh[q] = *reinterpret_cast<const float4 *>(shift + n0 + 4 * lhi + 8 * q);
            if (X3) { sc[q].x *= 1.0f / 2048.0f; sc[q].y *= 1.0f / 2048.0f; sc[q].z *= 1.0f / 2048.0f; sc[q].w *= 1.0f / 2048.0f; }
        }
    }
    const float lo = relu ? 0.0f : -__builtin_huge_valf();   // branch-free ReLU (this file is compiled with -fno-honor-nans)

    RF_SETUP_X(0)
    RF_ISSUE_X(0, 0)
    RF_ISSUE_X(1, 1)                                       // Cin >= 64: at least two chunks per tile
#pragma unroll
    for (int u = 0; u < NFA; ++u) RF_LOAD_A(u, fa[u]);
    // (waiting for the first patch only and letting the rest land behind counted waits was measured: no gain)
    SFD2_BARRIER_DRAIN();
    if constexpr (CPB != 0) {
        RF_READ_B8(Xs, 0, fb8[0]);
    } else {
        RF_READ_B(Xs, 0, 0, fb[0]);
        RF_READ_B(Xs, 0, 1, fb[1]);
    }

    int sb1 = 0x7f7f7f7f;                                  // the B side's scale bytes (2^0), in a register for the asm form
    asm volatile("" : "+v"(sb1));
    int bc = 0;                                            // C % 3: three patch buffers
    int c = 0, seq = 0, C = 0;                             // chunk within the tile, tile of this block, chunk of this block
    // one chunk; CC = the chunk's index within the tile as a constant (RES), or -1 (runtime c)
    auto chunk = [&](auto cc_tag) {
        constexpr int CC = decltype(cc_tag)::value;
        // a corr-plane chunk (COMP & 1)?  A wave-uniform RUNTIME flag: two instantiations of this body (one per chunk type) cost
        // 70 more registers than one (each copy keeps its own hoisted state across the shared rings) and spill in the unit loop
        const bool F8 = (CPB & 1) && c >= NCP;
        const int f8s = __builtin_amdgcn_readfirstlane(F8 ? 1 : 0);   // (a scalar register for the branch inside sfd2_mfma_unit2)
        if (X3 && c == NCP) {                               // the hi x hi sums are complete: the cross terms carry 2^-11
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] *= 2048.0f;
        }
        const int bn = bc == 2 ? 0 : bc + 1, bnn = bn == 2 ? 0 : bn + 1;
        const unsigned char *xs = Xs + bc * G::XBYTES;
        const unsigned char *xn = Xs + bn * G::XBYTES;
        // the patch two chunks ahead (of this or of the next tile) goes into the buffer chunk C - 1 used (every wave passed
        // that chunk's boundary barrier with its reads retired), one piece per unit so that the copies of all blocks do not
        // arrive at the memory system as one burst.  The block's last two chunks re-request its last patch (branch-free:
        // the in-flight counts stay what the waits assume).
        const int Cx = C + 2 < TC ? C + 2 : TC - 1;
        const int seq_x = Cx / NCH, cx = Cx - seq_x * NCH;
        if constexpr (CPB == 0) {   // (the compensated forms have two chunk bodies: their caller does this once)
            if (seq_x != xoff_seq) RF_SETUP_X(seq_x)
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t < G::PPW && !(ABL & 4)) RF_ISSUE_X1(cx, bnn, t);
            if (t == 8) {
                // chunk boundary: my pieces of the next patch have landed (VMW younger operations may still be in flight;
                // a tile's epilogue only adds younger ones), my reads of this one are done; then the whole block's.  The
                // last chunk's look-ahead reads return stale records that no MFMA consumes.  (RES issues no filter
                // loads: only the PPW pieces of the patch after next are younger.)
                static_assert(RfGeom<1>::VMW == 33 && RfGeom<2>::VMW == 30, "chunk-boundary wait counts");
                if (RES) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if (S == 2) asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(33) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CPB != 0) {
                // the pixel fragments of unit t + 1 (of the next chunk's first unit at t = 8: its patch has landed, see above)
                if (!(ABL & 2)) {
                    if (t + 1 < 9) RF_READ_B8(xs, t + 1, fb8[(t + 1) & 1]);
                    else RF_READ_B8(xn, 0, fb8[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 16)) {
                    // both chunk types are ONE statement to hipcc (a scalar branch inside the asm picks four fp16 or two fp8 MFMAs): with an
                    // if / else around builtins it copied every accumulator back to its home registers where the paths meet -- 160 v_mov_b64 per
                    // chunk, as much VALU time as the chunk's MFMAs (profiles/r04_conv2b_ablations.txt)
                    static_assert(NF == 2, "sfd2_mfma_unit2: two pixel fragments per wave");
                    h8_t x0 = fa[t][0], x1 = fa[t][1];
                    asm volatile("" : "+v"(x0), "+v"(x1));
                    sfd2_mfma_unit2(acc[0], acc[1], x0, x1, sfd2_cat8(x0, x1), fb8[t & 1][0], fb8[t & 1][1], sa, sb1, f8s);
                }
                __builtin_amdgcn_sched_barrier(0);
                int ua = c * 9 + t + 9;
                if (ua >= NU) ua -= NU;
                if (!(ABL & 1)) RF_LOAD_A(ua, fa[t]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int h = 2 * t + kk;                  // K half of the chunk; its fragments sit in fb[h % 3]
                if (ABL & 2) {
                } else if (h + 2 < 18) RF_READ_B(xs, (h + 2) / 2, (h + 2) % 2, fb[(h + 2) % 3]);
                else RF_READ_B(xn, (h + 2 - 18) / 2, (h + 2) % 2, fb[(h + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[RES ? CC * 9 + t : t][kk], fb[h % 3][f], acc[f], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!RES) {
                // nine units ahead: the same filters serve every tile, so the ring wraps at a tile's last unit
                int ua = c * 9 + t + 9;
                if (ua >= NU) ua -= NU;
                if (!(ABL & 1)) RF_LOAD_A(ua, fa[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (CPB != 0) {
            // nine units per chunk: the next chunk's first fragments were read into slot 1; its units index from slot 0
#pragma unroll
            for (int f = 0; f < NF; ++f) fb8[0][f] = fb8[1][f];
        }
        bc = bn;
        ++C;
    };
    // a tile's epilogue: y = acc * scale + shift (ReLU), regrouped with v_permlane32_swap into 16-byte stores
    // (conv2_kernels.hip); the next tile's first operands are already in registers / in flight
    unsigned int smax = 0;         // range status: the largest value in front of the saturation, wave-uniform across the tiles
    auto epilogue = [&]() {
        float mx = 0.0f;
        if constexpr ((CPB & 1) != 0) sfd2_mfma_settle();   // the tile's last MFMAs were issued by inline asm
        if (SS_RELOAD) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sc[q] = *reinterpret_cast<const float4 *>(scale + n0 + 4 * lhi + 8 * q);
                sh[q] = *reinterpret_cast<const float4 *>(shift + n0 + 4 * lhi + 8 * q);
            }
        }
        const int swz_ = xcd_swizzle4((int)blockIdx.x + seq * (int)gridDim.x, n_tiles);
        const int tx_ = swz_ % tiles_x, ty_ = swz_ / tiles_x;
        const int oy0 = ty_ * RF_TH, ox0 = tx_ * RF_TW;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int oy = oy0 + 2 * (f0 + f) + ly, ox = ox0 + lx;
            const bool inb = oy < Ho && ox < Wo;
            const size_t pix = (size_t)(inb ? oy : 0) * Wo + (inb ? ox : 0);
            if constexpr ((COMP & 64) != 0) {
                // fp6 corr records (sfd2_epi16_fp6, sfd2_internal.h): the fragment's 16 channels of this lane are one half-record of its pixel
                uint2 hv4[4];
                uint4 r0, r1;
                sfd2_epi16_fp6(acc[f], sc, sh, relu ? 0.0f : -SFD2_C_SAT, hv4, r0, r1, mx, inb);
                const size_t ob = pix * CoutP + n0;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const auto t0 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].x, hv4[2 * m + 1].x, false, false);
                    const auto t1 = __builtin_amdgcn_permlane32_swap(hv4[2 * m].y, hv4[2 * m + 1].y, false, false);
                    if (inb) *reinterpret_cast<uint4 *>(out + ob + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                }
                if (inb) {
                    *reinterpret_cast<uint4 *>(out_c + ob + 8 * lhi) = r0;
                    *reinterpret_cast<uint4 *>(out_c + ob + 16 + 8 * lhi) = r1;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
                continue;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const size_t o16 = pix * CoutP + n0 + 8 * (2 * m + lhi);
                uint2 pk[2], ck[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    if constexpr ((CPB & 2) != 0) {
                        sfd2_epi4<false>(acc[f][4 * q + 0], acc[f][4 * q + 1], acc[f][4 * q + 2], acc[f][4 * q + 3], sc[q], sh[q], sc[q],
                                         relu ? 0.0f : -SFD2_C_SAT, pk[j], ck[j], mx, inb);
                    } else if constexpr (X3) {
                        float v0 = acc[f][4 * q + 0] * sc[q].x + sh[q].x;
                        float v1 = acc[f][4 * q + 1] * sc[q].y + sh[q].y;
                        float v2 = acc[f][4 * q + 2] * sc[q].z + sh[q].z;
                        float v3 = acc[f][4 * q + 3] * sc[q].w + sh[q].w;
                        v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
                        if constexpr ((COMP & 8) != 0) {
                            if (inb) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(out) + pix * CoutP + n0 + 4 * lhi + 8 * q) = make_float4(v0, v1, v2, v3);
                        } else {
                            const h4_t hv = cvt4r(v0, v1, v2, v3);
                            const h4_t lv = cvt4r((v0 - (float)hv[0]) * 2048.0f, (v1 - (float)hv[1]) * 2048.0f, (v2 - (float)hv[2]) * 2048.0f,
                                                  (v3 - (float)hv[3]) * 2048.0f);
                            __builtin_memcpy(&pk[j], &hv, 8);
                            __builtin_memcpy(&ck[j], &lv, 8);
                        }
                    } else {
                        float v0 = acc[f][4 * q + 0] * sc[q].x + sh[q].x;
                        float v1 = acc[f][4 * q + 1] * sc[q].y + sh[q].y;
                        float v2 = acc[f][4 * q + 2] * sc[q].z + sh[q].z;
                        float v3 = acc[f][4 * q + 3] * sc[q].w + sh[q].w;
                        v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); v2 = fmaxf(v2, lo); v3 = fmaxf(v3, lo);
                        const h4_t hv = cvt4r(v0, v1, v2, v3);
                        __builtin_memcpy(&pk[j], &hv, 8);
                    }
                }
                if constexpr (X3 && (COMP & 8) != 0) continue;      // fp32 rows went out above
                if constexpr (OUTC) {
                    const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                    const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                    if (inb) *reinterpret_cast<uint4 *>(out_c + o16) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                }
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1 = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                const uint4 v16 = make_uint4(t0[0], t1[0], t0[1], t1[1]);
                if (STG) {
                    // through LDS: [128 pixels][BN channels] fp16 with the 16-byte slots XOR-swizzled by the pixel index
                    const int px = 32 * (f0 + f) + l32;
                    const int slot = (n0 >> 3) + 2 * m + lhi;
                    *reinterpret_cast<uint4 *>(Stage + px * (BN * 2) + ((slot ^ (px & 15)) << 4)) = v16;
                } else if (inb && (!(ABL & 8) || t0[0] == 0x12345678u)) {
                    *reinterpret_cast<uint4 *>(out + o16) = v16;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;
        }
        if (STG) {
            // ... and out again as whole pixel rows: a store instruction covers 64 consecutive 16-byte slots = four
            // neighbouring pixels x 256 B = 1 KB of contiguous output (the direct form writes 32 B of 32 pixels per
            // instruction; the output of conv2a is 123 MB and its stores were as long as its MFMAs)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            constexpr int SLOTS = BN / 8;                  // 16-byte slots per pixel
#pragma unroll
            for (int k = 0; k < 128 * SLOTS / 512; ++k) {
                const int i = tid + k * 512;
                const int px = i / SLOTS, slot = i - px * SLOTS;
                const int fr = px >> 5, l = px & 31;
                const int oy = oy0 + 2 * fr + (l >> 4), ox = ox0 + (l & 15);
                const uint4 v16 = *reinterpret_cast<const uint4 *>(Stage + px * (BN * 2) + ((slot ^ (px & 15)) << 4));
                if (oy < Ho && ox < Wo && !(ABL & 8))
                    *reinterpret_cast<uint4 *>(out + ((size_t)oy * Wo + ox) * CoutP + slot * 8) = v16;
            }
        }
        if constexpr ((CPB & 2) != 0) { const unsigned int wb = sfd2_wave_max_bits(mx); smax = wb > smax ? wb : smax; }
        c = 0;
        ++seq;
    };
    if constexpr (RES) {
        for (int s2 = 0; s2 < n_my; ++s2) {
            chunk(std::integral_constant<int, 0>{});
            chunk(std::integral_constant<int, 1>{});
            epilogue();
        }
    } else if constexpr ((CPB & 1) != 0) {
        while (C < TC) {
            {
                const int Cx = C + 2 < TC ? C + 2 : TC - 1;
                const int seq_x = Cx / NCH;
                if (seq_x != xoff_seq) RF_SETUP_X(seq_x)
            }
            chunk(std::integral_constant<int, -1>{});
            if (++c == NCH) epilogue();
        }
    } else {
        while (C < TC) {
            chunk(std::integral_constant<int, -1>{});
            if (++c == NCH) epilogue();
        }
    }
    if constexpr ((CPB & 2) != 0) sfd2_range_commit(range, smax);
#undef RF_ISSUE_X
#undef RF_ISSUE_X1
#undef RF_LOAD_A
#undef RF_READ_B
#undef RF_LOAD_A8
#undef RF_READ_B8
#undef RF_SETUP_X
}

template <int S, int BN = RF_BN, int ABL = 0, bool RES = false, int COMP = 0>
static void launch_rf_t(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                        const float *scale, const float *shift, int CoutP, int relu, half_t *out,
                        int Ho, int Wo, const half_t *zero_page, const half_t *in_c = nullptr, half_t *out_c = nullptr, int sa = 0,
                        unsigned int *range = nullptr)
{
    constexpr size_t lds = (size_t)3 * RfGeom<S>::XBYTES;
    static bool attr_done = false;
    static int slots = 256;
    auto kern = conv3x3_rf_kernel<S, BN, ABL, RES, COMP>;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;                                   // one resident block per CU (3 x 43 KB of LDS for stride 2)
        attr_done = true;
    }
    const int tiles_x = (Wo + RF_TW - 1) / RF_TW, tiles_y = (Ho + RF_TH - 1) / RF_TH;
    const int n_tiles = tiles_x * tiles_y;                 // CoutP == BN: one channel tile
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out,
                       Ho, Wo, tiles_x, n_tiles, zero_page, in_c, out_c, sa, range);
}

// compensated instantiations (SFD2_PREC_F16C): the stride-2 layer with 128 output channels (conv2b).  wpk = the layer's wc
// array (32-wide chunks, hi then corr), sbyte its scale byte.  false = no instantiation for this shape.
// the geometries launch_conv3x3_rf_c takes (a caller that wants fp6 output records has to know beforehand)
bool conv3x3_rf_c_serves(int ks, int stride, int CoutP, int Cin, int Ho, int Wo)
{
    if (ks != 3 || Cin % 64 != 0 || !(CoutP == 128 && stride == 2)) return false;
    return (long long)(Ho * stride + 2) * (Wo * stride + 2) * Cin * (long long)sizeof(half_t) < (1ll << 31);
}
bool launch_conv3x3_rf_c(hipStream_t st, const half_t *in, const half_t *in_c, int H, int W, int Cin, const half_t *wpk,
                         const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out, half_t *out_c,
                         int Ho, int Wo, const half_t *zero_page, int sbyte, unsigned int *range, int fmt6)
{
    if (!in_c || !out_c || Cin % 64 != 0) return false;
    if ((long long)(Ho * stride + 2) * (Wo * stride + 2) * Cin * (long long)sizeof(half_t) >= (1ll << 31)) return false;
    const int sa = (sbyte & 255) * 0x01010101;
    if (CoutP == 128 && stride == 2) {
#ifdef SFD2_EXPERIMENTS
        if (const char *ab = sfd2_env("SFD2_RFC_ABL")) {
            switch (atoi(ab)) {
#define RFC_ABL_CASE(a_) case a_: launch_rf_t<2, 128, a_, false, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range); return true;
                RFC_ABL_CASE(1) RFC_ABL_CASE(2) RFC_ABL_CASE(4) RFC_ABL_CASE(8) RFC_ABL_CASE(16) RFC_ABL_CASE(5) RFC_ABL_CASE(7) RFC_ABL_CASE(23)
#undef RFC_ABL_CASE
            default: break;
            }
        }
#endif
        if (fmt6 & 2) launch_rf_t<2, 128, 0, false, 3 | 64>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        else launch_rf_t<2, 128, 0, false, 3>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page, in_c, out_c, sa, range);
        return true;
    }
    return false;
}

// SFD2_PREC_F16X3 on this kernel (256-channel tiles): in_hi / in_lo planes, wpl = the filters as hi units then lo' units (the fp32 set's
// packing split by x3_split_planes), output as planes (out_hi / out_lo) or as fp32 [P][CoutP] (out_f32).  false = no instantiation.
bool launch_conv3x3_rf_x3(hipStream_t st, const half_t *in_hi, const half_t *in_lo, int H, int W, int Cin, const half_t *wpl,
                          const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out_hi, half_t *out_lo,
                          float *out_f32, int Ho, int Wo, const half_t *zero_page)
{
    if (Cin % 64 != 0 || (stride != 1 && stride != 2)) return false;
    if ((long long)(Ho * stride + 2) * (Wo * stride + 2) * Cin * (long long)sizeof(half_t) >= (1ll << 31)) return false;
    if (CoutP == 128 && stride == 2) {      // conv2b
        if (out_f32) launch_rf_t<2, 128, 0, false, 12>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, reinterpret_cast<half_t *>(out_f32), Ho, Wo, zero_page, in_lo, nullptr, 0);
        else launch_rf_t<2, 128, 0, false, 4>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, out_hi, Ho, Wo, zero_page, in_lo, out_lo, 0);
        return true;
    }
    if (CoutP != RF_BN) return false;
    if (out_f32) {
        half_t *o = reinterpret_cast<half_t *>(out_f32);
        if (stride == 2) launch_rf_t<2, RF_BN, 0, false, 12>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, o, Ho, Wo, zero_page, in_lo, nullptr, 0);
        else launch_rf_t<1, RF_BN, 0, false, 12>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, o, Ho, Wo, zero_page, in_lo, nullptr, 0);
    } else {
        if (stride == 2) launch_rf_t<2, RF_BN, 0, false, 4>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, out_hi, Ho, Wo, zero_page, in_lo, out_lo, 0);
        else launch_rf_t<1, RF_BN, 0, false, 4>(st, in_hi, H, W, Cin, wpl, scale, shift, CoutP, relu, out_hi, Ho, Wo, zero_page, in_lo, out_lo, 0);
    }
    return true;
}

// does conv3x3_rf serve this layer?  Shape AND output size decide (the filter packing is the 32-channel-chunk one that
// conv3x3_pp and the stride-2 conv_igemm2 use, so the choice can be made per launch).
// conv2a (64 -> 128 channels, stride 1) goes through the resident-filter variant at every size: its filters are packed in
// 32-channel chunks for that (conv_igemm2_chunk asks)
bool conv3x3_rf_resident(int ks, int stride, int CoutP, int Cin)
{
    static const bool no_res = sfd2_env("SFD2_RF_NO_RES") != nullptr;   // experiment: conv_igemm2 for conv2a
    return !no_res && ks == 3 && stride == 1 && CoutP == 128 && Cin == 64;
}

bool conv3x3_rf_serves(int ks, int stride, int CoutP, int Cin, int Ho, int Wo)
{
    static const char *mode = sfd2_env("SFD2_CONV_RF");   // experiments: "off", "all"
    if (mode && mode[0] == 'o') return false;
    if (ks != 3 || (stride != 1 && stride != 2) || Cin % 64 != 0) return false;
    // the patch copies address the input through a buffer descriptor with 32-bit byte offsets
    if ((long long)(Ho * stride + 2) * (Wo * stride + 2) * Cin * (long long)sizeof(half_t) >= (1ll << 31)) return false;
    if (conv3x3_rf_resident(ks, stride, CoutP, Cin)) return true;
    if (CoutP != RF_BN && !(CoutP == 128 && stride == 2)) return false;   // one channel tile per block: 256, or conv2b's 128
    if (mode && mode[0] == 'a') return true;
    if (stride == 2) return true;
    // stride 1: conv3x3_pp (512 pixels x 128 channels per block) is the faster kernel per FLOP but needs ~2 blocks per CU
    // worth of output; compare rounds on the 256 CUs weighted by the measured time of one round of each (Cin = 256:
    // ~60 us against ~37 us; 1600x1200: convPa.3 52 -> 40 us here, conv3b 113 -> 130 us)
    const long long pp_blocks = (long long)((Wo + 31) / 32) * ((Ho + 15) / 16) * (CoutP / 128);
    const long long rf_blocks = (long long)((Wo + RF_TW - 1) / RF_TW) * ((Ho + RF_TH - 1) / RF_TH) * (CoutP / RF_BN);
    return ((rf_blocks + 255) / 256) * 10 < ((pp_blocks + 255) / 256) * 16;
}

void launch_conv3x3_rf(hipStream_t st, const half_t *in, int H, int W, int Cin, const half_t *wpk,
                       const float *scale, const float *shift, int CoutP, int stride, int relu, half_t *out,
                       int Ho, int Wo, const half_t *zero_page)
{
#ifdef SFD2_EXPERIMENTS
    if (const char *ab = sfd2_env("SFD2_RF_ABL"); ab && CoutP == RF_BN) {
        switch (atoi(ab) * 4 + stride) {
#define RF_ABL_CASE(a_) \
        case (a_) * 4 + 1: launch_rf_t<1, RF_BN, a_>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page); return; \
        case (a_) * 4 + 2: launch_rf_t<2, RF_BN, a_>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page); return;
            RF_ABL_CASE(1) RF_ABL_CASE(2) RF_ABL_CASE(3) RF_ABL_CASE(4) RF_ABL_CASE(5) RF_ABL_CASE(6) RF_ABL_CASE(7) RF_ABL_CASE(8) RF_ABL_CASE(15)
#undef RF_ABL_CASE
        default: break;
        }
    }
#endif
    if (CoutP == 128 && stride == 1) launch_rf_t<1, 128, 0, true>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
    else if (CoutP == 128) launch_rf_t<2, 128>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
    else if (stride == 2) launch_rf_t<2>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
    else launch_rf_t<1>(st, in, H, W, Cin, wpk, scale, shift, CoutP, relu, out, Ho, Wo, zero_page);
}
