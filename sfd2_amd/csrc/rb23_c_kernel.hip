// SFD2_PREC_F16C, option "rb_inner" = 2: ResBlock.conv2 (3x3, groups = 32) + BN + ReLU and ResBlock.conv3 (1x1) + BN + residual +
// ReLU in ONE kernel (nets/sfd2.py:14-18, :32-55).  With t1 and t2 plain fp16 the grouped conv's output tile of 4 x 32 pixels x 256
// channels is 64 KB: it stays in LDS, in the record layout conv1x1_c256_c_kernel stages its input in, and never travels to
// HBM (2 x 61 MB per block at 1600x1200) -- and the block is one launch shorter.
//
//   phase G  gconv_c_kernel<false, false>'s arithmetic: 64-channel chunks of the 6 x 34 patch of t1 copied into an LDS ring
//            (one barrier per chunk), two groups per 16x16x32 MFMA with block-diagonal filter
//            fragments, second fp16 pass with the filter residuals; a wave = one group pair of the chunk x two tile rows
//   phase C  conv1x1_c256_c_kernel<true, false, true>'s arithmetic on the tile's four 32-pixel rows: a wave owns 32 output
//            channels, filters (fp16 + fp16 residuals, fragment order) in 128 registers for the life of the block
//
// Persistent, one block per CU; the same operations in the same order as the two kernels it replaces: results are identical
// bit for bit (tests/test_gpu_f16c.py).
#include "sfd2_internal.h"
#include <type_traits>

#define R23_NT 512
#ifndef R23_RES_AT
#define R23_RES_AT 3        // the chunk at whose top an L8 instantiation requests row 0's residual
#endif
#define R23_TH 4
#define R23_TW 32
#define R23_PH 6
#define R23_PW 34
#define R23_NPIX (R23_PH * R23_PW)                  // 204 patch pixels x 128 B (one 64-channel chunk) = 25.5 KB
#define R23_XB (26 * 1024)                          // a patch buffer: 26 copy instructions of 1 KB (the last one half used)
#define R23_T2B (R23_TH * R23_TW * 512)
#ifdef SFD2_RB23_OLDSWZ
#define R23_SWZ(c_) (((c_) >> 1) & 7)
#else
#define R23_SWZ(c_) ((((c_) >> 1) & 3) << 1)
#endif

typedef float r23_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void r23_lds_t;
typedef const __attribute__((address_space(1))) void r23_gbl_t;

__device__ __forceinline__ int r23_xcd_swizzle(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}

// -DSFD2_RB23_TRACE: cycle stamps of block 3's waves 0 and 7 (chunk tops, phase C start, rows, tile end) and the wall-clock exit
// of every block, printed by the launcher
#ifdef SFD2_RB23_TRACE
#include <stdio.h>
__device__ unsigned long long g_r23_cyc[2][6][12];
__device__ unsigned long long g_r23_wall[1024][2];
#define R23_CYC(k_) if (blockIdx.x == 3 && (wave == 0 || wave == 7) && lane == 0 && tcount < 6) g_r23_cyc[wave == 7][tcount][k_] = __builtin_readcyclecounter();
#define R23_WALL(k_) if (tid == 0) g_r23_wall[blockIdx.x][k_] = __builtin_amdgcn_s_memrealtime();
#else
#define R23_CYC(k_)
#define R23_WALL(k_)
#endif

// Pipeline.  Patch chunks travel global -> LDS by direct copies (global_load_lds), TWO chunks ahead of their use, into a ring of
// three buffers (a chunk's compute, ~1.5k cycles, is shorter than an HBM round trip); a 128-byte pixel record's eight 16-byte
// parts sit at slot part ^ R23_SWZ(patch column), R23_SWZ(c) = ((c >> 1) & 3) << 1.  What has to differ: one LDS cycle of a `ds_read_b128`
// serves sixteen lanes (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, ...) = columns {0-3, 12-15} of the lanes with g = 0 and columns 4-11 of
// those with g = 1 (the pair's other group: part + 1), shifted by the tap's kx.  Two columns of one parity share their ((c >> 1) & 3) term
// exactly when they are 8 apart -- one of them is then a g = 0 lane and the other a g = 1 lane whatever kx is, and the part bit keeps them
// apart; bit 0 of the slot is the part's, never the swizzle's.  (Until the end of round 5 the term was (c >> 1) & 7: conflict-free for
// kx = 0 only, 1.7 LDS cycles per lane group on average over a chunk's reads -- 26 % of the kernel's LDS cycles were conflicts,
// profiles/r05s_pmc_summary.txt; -DSFD2_RB23_OLDSWZ builds it.)  The row not entering the swizzle, a lane's 20 fragment addresses of a chunk are five per-lane
// offsets (one per K step: the tap differs between lane groups) plus compile-time (row, pixel half) offsets that fit the
// ds_read immediate.  A chunk is bound by VALU issue (16-wide SIMDs: a wave64 instruction takes four cycles, 40 MFMAs per chunk
// take 640), so what is computed per chunk is kept to the accumulator fold, the BN epilogue and a few adds.  The grouped conv's filter fragments of the NEXT chunk are loaded into the registers of the step that has
// just issued its MFMAs; the copies for chunk n + 2 go out after chunk n's MFMAs, i.e. BEHIND those loads in the wave's
// memory queue, so `vmcnt(4)` at the top of a chunk (every wave issues exactly four copies per chunk) covers the filters and
// the chunk's own patch and leaves the newest copies in flight.  Phase C: a row's residual is loaded two rows ahead (rows 0 / 1
// around chunk 3's copies, rows 2 / 3 during the epilogues of rows 0 / 1, into the registers those have just emptied), the next
// tile's first filter fragments before the last row's stores; those four stores are all that is still in flight at the top of
// the next tile.
// RES_R1 / OUT_R1 (round 4, option "trunk_r1"): the block's input / output carries ONE correction byte per channel -- the residual
// e4m3((x - hi) * 2^9), 256 bytes per pixel at the start of the corr plane -- instead of the (residual, value) unit: the tensors between the
// ResBlocks are read by conv1x1_c256_c<.., 2, ..> (which takes the value term from the hi plane) and by this kernel's skip path (which never
// used the value byte), and these kernels are bound by bytes: 3 instead of 4 per channel.
template <bool RES_R1, bool OUT_R1>
__global__ __launch_bounds__(R23_NT, 2)
void rb23_c_kernel(const half_t *__restrict__ t1, int H, int W,
                   const half_t *__restrict__ w2h /*[16 pairs][5 steps][64 lanes][8]*/, const half_t *__restrict__ w2l /*residuals * 2^11, same layout*/,
                   const float *__restrict__ sc2, const float *__restrict__ sh2,
                   const half_t *__restrict__ w3h /*fragment order [8 waves][8][64 lanes][16]*/, const half_t *__restrict__ w3l,
                   const float *__restrict__ sc3, const float *__restrict__ sh3,
                   const half_t *__restrict__ res, const half_t *__restrict__ res_c,
                   half_t *__restrict__ out, half_t *__restrict__ out_c, int tiles_x, int n_tiles,
                   unsigned int *__restrict__ range_t2, unsigned int *__restrict__ range_out /* range-status slots of t2 (LDS-resident) and of the block's output, or null */,
                   int sa3 /* RES_R1: conv3's corr scale byte x 0x01010101; then w3l = its filter residuals as e4m3, [8 waves][4][64 lanes][32 B] */)
{
    // RES_R1 instantiations (the default path): conv3's term x * lo_w runs on the scaled MFMA -- t2's value bytes e4m3(t2 / 4), rebuilt from the
    // lane's fp16 fragments of four K steps (sixteen v_cvt_scalef32_pk_fp8_f16), against the filter residuals as e4m3 -- instead of a second
    // fp16 pass: 4 x 66 cycles instead of 16 x 32 per row, 32 resident registers instead of 64 and no second accumulator.
    constexpr bool L8 = RES_R1;
    unsigned int smax_t2 = 0, smax_out = 0;                 // wave-uniform across the tiles
    // Three separate LDS objects, not slices of one array: the compiler orders every LDS store behind all pending direct-to-LDS
    // copies it cannot prove disjoint from it (`s_waitcnt vmcnt(0)` in front of the first T2 store of every chunk: the copies
    // just issued had to land before the epilogue went on -- no prefetch at all)
    __shared__ __attribute__((aligned(16))) unsigned char T2[R23_T2B];     // [128 pixels][512 B], 16-byte slots XOR (pixel & 31)
    __shared__ __attribute__((aligned(16))) unsigned char XP[3 * R23_XB + 1024];   // patch ring, then 1 KB that swallows the idle copies
    __shared__ __attribute__((aligned(16))) float SS[4 * 256];            // sc2, sh2, sc3, sh3

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, lcol = lane & 15;              // phase G
    const int pw = wave & 3, rh = wave >> 2;

    // conv3's filters: resident
    h8_t ah[16];
    v8i_t al[L8 ? 4 : 8];
    if constexpr (L8) {
        const size_t fo = ((size_t)wave * 8 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[2 * c] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16);
            ah[2 * c + 1] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16 + 8);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const half_t *p8 = w3l + ((size_t)(wave * 4 + c) * 64 + lane) * 16;
            al[c] = sfd2_cat8(*reinterpret_cast<const h8_t *>(p8), *reinterpret_cast<const h8_t *>(p8 + 8));
        }
    } else {
        const size_t fo = ((size_t)wave * 8 * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            ah[2 * c] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16);
            ah[2 * c + 1] = *reinterpret_cast<const h8_t *>(w3h + fo + (size_t)c * 64 * 16 + 8);
            al[c] = sfd2_cat8(*reinterpret_cast<const h8_t *>(w3l + fo + (size_t)c * 64 * 16), *reinterpret_cast<const h8_t *>(w3l + fo + (size_t)c * 64 * 16 + 8));
        }
    }
    for (int t = tid; t < 256; t += R23_NT) { SS[t] = sc2[t]; SS[256 + t] = sh2[t]; SS[512 + t] = sc3[t]; SS[768 + t] = sh3[t]; }

    // copies of one chunk: instruction j = wave + 8 * i (i < 4) moves pieces j * 64 + lane (pixel = piece >> 3, slot = piece & 7);
    // j >= 26 has nothing to move -- every wave issues four all the same, the counted waits rely on it.  `buffer_load ... lds` with
    // per-lane BYTE OFFSETS computed once per tile (R23_SETUP): padding, pixels outside the image and the idle instructions get
    // an offset beyond the buffer's range, for which the hardware returns zeros; a chunk is the scalar offset chunk * 128.
    // (As `global_load ... lds` with the address arithmetic and the bounds tests per copy, a chunk's four copies were ~240 of
    // its ~760 instructions, and the chunk is bound by instruction issue: 40 MFMAs.)
    const int t1_bytes = (int)((size_t)H * W * 256 * sizeof(half_t));
    const auto t1_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(t1), 0, t1_bytes, 0x00020000);
    int xoff[4];
#define R23_SETUP(oy0_, ox0_)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        const int j = wave + 8 * i;                                                                       \
        int piece = j * 64 + lane;                                                                        \
        asm volatile("" : "+v"(piece));   /* (recomputed per tile) */                                      \
        const int q = piece >> 3;                                                                         \
        const int py = (q * 241) >> 13, px = q - py * R23_PW;                                             \
        const int iy = (oy0_)-1 + py, ix = (ox0_)-1 + px;                                                  \
        const int part = (piece & 7) ^ R23_SWZ(px);       /* slot = part ^ R23_SWZ(column) */                  \
        const bool ok = j < 26 && q < R23_NPIX && iy >= 0 && iy < H && ix >= 0 && ix < W;                  \
        xoff[i] = ok ? ((iy * W + ix) * 256 + part * 8) * (int)sizeof(half_t) : (int)0x80000000;           \
    }
#define R23_COPIES(chunk_, buf_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
        const int j = wave + 8 * i;                                                                       \
        unsigned char *dst = j < 26 ? XP + (buf_)*R23_XB + j * 1024 : XP + 3 * R23_XB;                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t1_rs, (r23_lds_t *)dst, 16, xoff[i], (chunk_)*128, 0, 0); \
    }

    // Filter-fragment reloads are issued by hand: a load the compiler tracks makes it wait for `vmcnt(0)` in front of the first
    // MFMA of the next chunk (it cannot count across the loop edge), i.e. for the copies issued behind the reload as well.  The
    // counted wait at the top of the chunk carries the registers as operands, so nothing that reads them moves above it.
#define R23_WLOAD(dst_, ptr_) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst_) : "v"(ptr_) : "memory")
#define R23_WLOAD_S(dst_, voff_, sptr_) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst_) : "v"(voff_), "s"(sptr_) : "memory")
#define R23_WTIE() asm volatile("" : "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]), "+v"(wh[4]), "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]), "+v"(wl[4]))

    // phase G, per lane and K step s: byte offset of the fragment of (tile row 0, pixel half 0) in a patch buffer
    int boff[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        int tap = 2 * s + (g >> 1);
        if (tap > 8) tap = 8;                               // zero-weight slot: read any valid location
        const int ky = tap / 3, kx = tap - ky * 3;
        const int col = lcol + kx;                          // (+ 16 for the second pixel half: the swizzle term does not change)
        boff[s] = ((rh * 2 + ky) * R23_PW + col) * 128 + (((pw * 2 + (g & 1)) ^ R23_SWZ(col)) << 4);
    }
    // T2 store of (tile row rh * 2, pixel half h): pixel pl = rh * 64 + 16 h + lcol, slot (pair * 2 + (g >> 1)) ^ (pl & 31) -- the
    // pair's chunk bits (chunk * 8) and the half's bit 4 enter by XOR
    const int t2b = (rh * 64 + lcol) * 512 + (g & 1) * 8;
    const int t2s = (pw * 2 + (g >> 1)) ^ lcol;             // slot for chunk 0, half 0
    const int wro = lane * 16;                              // filter fragments: byte offset of this lane in a 1 KB fragment

    int tile = blockIdx.x;
    int oy0, ox0;
    {
        const int swz = r23_xcd_swizzle(tile, n_tiles);
        oy0 = (swz / tiles_x) * R23_TH; ox0 = (swz % tiles_x) * R23_TW;
    }
    h8_t wh[5], wl[5];                                      // grouped conv: fragments of the chunk at hand / being loaded for the next one
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        R23_WLOAD(wh[s], w2h + ((size_t)(pw * 5 + s) * 64 + lane) * 8);
        R23_WLOAD(wl[s], w2l + ((size_t)(pw * 5 + s) * 64 + lane) * 8);
    }
    // (conv3's filters are waited for HERE: a load still pending at the head of the tile loop makes the compiler drain the
    //  memory queue in front of the chunk loop of every tile)
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(ah[2 * c]), "+v"(ah[2 * c + 1]), "+v"(al[L8 ? c & 3 : c]));
    R23_SETUP(oy0, ox0)
    R23_COPIES(0, 0)
    R23_COPIES(1, 1)
    int tcount = 0;
    (void)tcount;
    R23_WALL(0)
    int ring = 0;                                           // buffer of the chunk at hand; the copies go to (ring + 2) % 3
    bool drain = true;                                      // top of a tile: full wait; the chunks behind it wait by count

    using rc_t = typename std::conditional<RES_R1, uint2, uint4>::type;
    uint4 rqa[2], rqb[2];                   // residuals of two rows in flight (even / odd rows)
    rc_t rca[2], rcb[2];
#define R23_RES(r4_, rq_, rc_)                                                                                      \
        {                                                                                                 \
            /* (rows / columns past the image repeat its last row / column: same values, same addresses, no predicate) */ \
            const int oy_ = oy0 + (r4_) < H ? oy0 + (r4_) : H - 1, ox_ = ox0 + lrow < W ? ox0 + lrow : W - 1; \
            const size_t ob_ = (size_t)(oy_ * W + ox_) * 256 + wave * 32;                                 \
            _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                               \
                rq_[m] = *reinterpret_cast<const uint4 *>(res + ob_ + 8 * (2 * m + lhi));                  \
                if constexpr (RES_R1) rc_[m] = *reinterpret_cast<const rc_t *>(reinterpret_cast<const unsigned char *>(res_c) + ob_ + 8 * (2 * m + lhi)); \
                else rc_[m] = *reinterpret_cast<const rc_t *>(res_c + ob_ + 8 * (2 * m + lhi));            \
            }                                                                                             \
        }
    for (;;) {
        const int next = tile + (int)gridDim.x;
        const bool has_next = next < n_tiles;
        int noy0 = oy0, nox0 = ox0;
        if (has_next) {
            const int swz = r23_xcd_swizzle(next, n_tiles);
            noy0 = (swz / tiles_x) * R23_TH; nox0 = (swz % tiles_x) * R23_TW;
        }
        // (range status: reduced per chunk / per row into the scalars -- a vector register carried through the chunk loop spills here)
        // ------------------------------------------------ phase G: grouped 3x3 -> T2
#pragma unroll 1
        for (int chunk = 0; chunk < 4; ++chunk) {
            // this chunk's patch (copied two chunks ago) and filter fragments have landed; the newest four copies may still fly.
            // One barrier per chunk: the buffer the copies below go to was last read a chunk ago, T2 in the previous tile's phase C
#ifdef SFD2_RB23_TRACE
            if (chunk == 1) { R23_CYC(10) }
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            if (chunk == 1) { R23_CYC(11) }
            asm volatile("s_barrier" ::: "memory");
#else
            if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            R23_WTIE();
            R23_CYC(chunk)
            drain = false;
            if (L8 && chunk == R23_RES_AT) {
                // (L8 instantiations have the registers for it) row 0's residual is requested HERE, ahead of phase C: requested behind chunk 3 its
                // HBM round trip sat in front of row 0 (profiles/r04s_rb23_trace.txt, column "residual + copies + barrier": 4-6 k of a 43 k-cycle
                // tile).  Nothing else is issued between here and the chunk's counted wait that this would be counted into: the wait at the top of
                // the next chunk allows four newer operations -- the residual's four loads are older than those copies.
                int ln0 = lane;
                asm volatile("" : "+v"(ln0));
                const int lrow = ln0 & 31, lhi = ln0 >> 5;
                R23_RES(0, rqa, rca)
                // (row 1's residual here as well, or row 0's a chunk earlier: 71.6 / 70.0 / 74.1 us per block -> 75.0 / 72.3 / 73.8 and 74.0 / 72.7 / 77.0)
            }
            const unsigned char *Xb = XP + ring * R23_XB;
            const int pair = chunk * 4 + pw;
            r23_f4 acc[4], acl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[t] = (r23_f4){0.0f, 0.0f, 0.0f, 0.0f}; acl[t] = acc[t]; }
            const int npair = ((chunk + 1) & 3) * 4 + pw;    // the next chunk's pair (chunk 3: the next tile's chunk 0, loaded in phase C)
            const half_t *nwh = w2h + (size_t)npair * 5 * 512, *nwl = w2l + (size_t)npair * 5 * 512;   // wave-uniform
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const unsigned char *bp = Xb + boff[s];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const h8_t bh = *reinterpret_cast<const h8_t *>(bp + ((t >> 1) * R23_PW + (t & 1) * 16) * 128);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], bh, acc[t], 0, 0, 0);
                    acl[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], bh, acl[t], 0, 0, 0);
                }
#if !(defined(SFD2_RB23_ABL) && (SFD2_RB23_ABL & 1))   // timing ablation (wrong results): no filter reloads
                if (chunk < 3) {
                    R23_WLOAD_S(wh[s], wro, nwh + s * 512);
                    R23_WLOAD_S(wl[s], wro, nwl + s * 512);
                }
#endif
            }
            if (chunk < 3) {   // copies for the chunk after next (this tile's, or the next tile's chunk 0); chunk 3's: below
                const int b2 = ring >= 1 ? ring - 1 : 2;    // (ring + 2) % 3
                if (chunk == 2) { R23_SETUP(noy0, nox0) }   // from here on the copies are the next tile's (past the last tile: this tile's again, never read)
                R23_COPIES((chunk + 2) & 3, b2)
            }
            const int c0 = pair * 16 + g * 4;
            const float4 sc = sfd2_lds_f4(SS + c0);
            const float4 sh = sfd2_lds_f4(SS + 256 + c0);
            const unsigned t2c = (unsigned)(size_t)(const r23_lds_t *)T2 + t2b;
            const int slc = t2s ^ (chunk * 8);
            unsigned int mx_t2 = 0;       // (packed fp16 maxima of the stored words: sfd2_track_h4)
            float mx_none = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fmaf(acl[t][r], 1.0f / 2048.0f, acc[t][r]);
                uint2 hv, cv;
                sfd2_epi4<false, false>(acc[t][0], acc[t][1], acc[t][2], acc[t][3], sc, sh, sc, 0.0f, hv, cv, mx_none);
                sfd2_track_h4(hv, mx_t2);
                // (stored by hand: in front of a compiler-visible LDS store behind pending direct-to-LDS copies the compiler puts
                //  `s_waitcnt vmcnt(0)` -- it did here for one of the four stores although T2 is an object of its own -- i.e. the
                //  chunk's epilogue waited for the copies issued a moment before it: one memory round trip per chunk)
                const unsigned t2a = t2c + ((slc ^ ((t & 1) * 16)) << 4) + ((t >> 1) * 32 + (t & 1) * 16) * 512;
                asm volatile("ds_write_b64 %0, %1" ::"v"(t2a), "v"(hv) : "memory");
            }
            { const unsigned int wb = sfd2_wave_max_bits(sfd2_h2_max(mx_t2)); smax_t2 = wb > smax_t2 ? wb : smax_t2; }
            ring = ring == 2 ? 0 : ring + 1;
        }
        // ------------------------------------------------ phase C: 1x1 + residual over the tile's four rows
        R23_CYC(4)
        {
            int ln0 = lane;
            asm volatile("" : "+v"(ln0));                   // (recomputed: a lane-derived value carried through the chunk loop is a spill)
            const int lrow = ln0 & 31, lhi = ln0 >> 5;
            if (!L8) { R23_RES(0, rqa, rca) }               // row 0's residual IN FRONT of chunk 3's copies (below), row 1's behind them
            else { (void)lrow; (void)lhi; }
        }

        {
            const int b2 = ring == 2 ? 0 : ring + 1;        // chunk 3's buffer + 2 (ring has already moved on by one)
            R23_COPIES(1, b2)
        }
        {
            int ln0 = lane;
            asm volatile("" : "+v"(ln0));
            const int lrow = ln0 & 31, lhi = ln0 >> 5;
            R23_RES(1, rqb, rcb)
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // T2 complete
        R23_CYC(5)
        // one row of phase C; LAST: behind it comes the next tile (its first filter fragments are requested here, on EVERY path, so
        // that the registers are free through the rows before), otherwise the next row's residual
        auto row = [&](const int r4u, uint4 (&rq)[2], rc_t (&rc)[2]) __attribute__((always_inline)) {
            const bool last = r4u == 3;
            const int r4 = oy0 + r4u < H ? r4u : H - 1 - oy0;     // rows past the image repeat its last row (same values to the same place)
            int ln = lane;
            asm volatile("" : "+v"(ln));   // (as above)
            const int lhi = ln >> 5;
            const int lrow = ox0 + (ln & 31) < W ? (ln & 31) : W - 1 - ox0;   // columns past the image repeat its last column
            const int oy = oy0 + r4, ox = ox0 + lrow;
            const size_t obase = (size_t)(oy * W + ox) * 256 + wave * 32;
            f32x16_t acc, acl;
            unsigned int mx_out = 0;
            float mx_none = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; acl[r] = 0.0f; }
            int sw = lrow;
            asm volatile("" : "+v"(sw));   // (the 16 fragment addresses: per row, not hoisted)
            const unsigned char *xp = T2 + (r4 * 32 + sw) * 512;
            // -DSFD2_RB23_PRIO (round 5 experiment): priority 1 for a row's MFMA burst -- the rows have no barriers between them, so the two waves of a SIMD could
            // drift half a row apart (one in its MFMAs, one in its epilogue) as the matcher's do with the same switch
#ifdef SFD2_RB23_PRIO
            __builtin_amdgcn_s_setprio(1);
#endif
            if constexpr (L8) {
                typedef short s2v_t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v8i_t vb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int kk = 4 * j + i;
                        const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], b, acc, 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            s2v_t v = {0, 0};
                            v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(v, h2_t{b[4 * q], b[4 * q + 1]}, 4.0f, false);
                            v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(v, h2_t{b[4 * q + 2], b[4 * q + 3]}, 4.0f, true);
                            int d;
                            __builtin_memcpy(&d, &v, 4);
                            vb[2 * i + q] = d;
                        }
                    }
                    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al[j], vb, acc, 0, 0, 0, sa3, 0, 0x7f7f7f7f);
                }
                asm volatile("" : "+v"(acc));
#ifdef SFD2_RB23_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            } else
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const h8_t b = *reinterpret_cast<const h8_t *>(xp + (((kk * 2 + lhi) ^ sw) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], b, acc, 0, 0, 0);
                acl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sfd2_half8(al[kk >> 1], kk & 1), b, acl, 0, 0, 0);
            }
            if constexpr (!L8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = __builtin_fmaf(acl[r], 1.0f / 2048.0f, acc[r]);
            }

            uint2 rp[2][2], rcp[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(rq[m].x, rq[m].z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(rq[m].y, rq[m].w, false, false);
                rp[m][0] = make_uint2(s0[0], s1[0]); rp[m][1] = make_uint2(s0[1], s1[1]);
                if constexpr (RES_R1) {      // 8 residual bytes per lane: after the swap, dword j = the bytes of this lane's channels 8 (2 m + j) + 4 lhi ..
                    const auto c0 = __builtin_amdgcn_permlane32_swap(rc[m].x, rc[m].y, false, false);
                    rcp[m][0] = make_uint2(c0[0], 0u); rcp[m][1] = make_uint2(c0[1], 0u);
                } else {
                    const auto c0 = __builtin_amdgcn_permlane32_swap(rc[m].x, rc[m].z, false, false);
                    const auto c1 = __builtin_amdgcn_permlane32_swap(rc[m].y, rc[m].w, false, false);
                    rcp[m][0] = make_uint2(c0[0], c1[0]); rcp[m][1] = make_uint2(c0[1], c1[1]);
                }
            }
            if (r4u < 2) { R23_RES(r4u + 2, rq, rc) }       // two rows ahead, into the registers this row's residual has just left
            else if (last) {
#pragma unroll
                for (int s = 0; s < 5; ++s) {
                    R23_WLOAD_S(wh[s], wro, w2h + (size_t)(pw * 5 + s) * 512);
                    R23_WLOAD_S(wl[s], wro, w2l + (size_t)(pw * 5 + s) * 512);
                }
            }
            const int cl = wave * 32 + 4 * lhi;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uint2 pk[2], ck[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = 2 * m + j;
                    const float4 sc = sfd2_lds_f4(SS + 512 + cl + 8 * q);
                    const float4 sh = sfd2_lds_f4(SS + 768 + cl + 8 * q);
                    h4_t rr;
                    __builtin_memcpy(&rr, &rp[m][j], 8);
                    constexpr float ks = 1.0f / (float)(1 << SFD2_C_XL_SHIFT);
                    const float4 ad = RES_R1 ? make_float4((float)rr[0] + __builtin_amdgcn_cvt_f32_fp8((int)rcp[m][j].x, 0) * ks,
                                                           (float)rr[1] + __builtin_amdgcn_cvt_f32_fp8((int)rcp[m][j].x, 1) * ks,
                                                           (float)rr[2] + __builtin_amdgcn_cvt_f32_fp8((int)rcp[m][j].x, 2) * ks,
                                                           (float)rr[3] + __builtin_amdgcn_cvt_f32_fp8((int)rcp[m][j].x, 3) * ks)
                                             : make_float4((float)rr[0] + sfd2_corr_lo(rcp[m][j].x, 0), (float)rr[1] + sfd2_corr_lo(rcp[m][j].x, 1),
                                                           (float)rr[2] + sfd2_corr_lo(rcp[m][j].y, 0), (float)rr[3] + sfd2_corr_lo(rcp[m][j].y, 1));
                    if constexpr (OUT_R1) sfd2_epi4_r1<true, false>(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], sc, sh, ad, 0.0f, pk[j], ck[j].x, mx_none);
                    else sfd2_epi4<true, false>(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], sc, sh, ad, 0.0f, pk[j], ck[j], mx_none);
                    sfd2_track_h4(pk[j], mx_out);
                }
                const auto t0 = __builtin_amdgcn_permlane32_swap(pk[0].x, pk[1].x, false, false);
                const auto t1v = __builtin_amdgcn_permlane32_swap(pk[0].y, pk[1].y, false, false);
                *reinterpret_cast<uint4 *>(out + obase + 8 * (2 * m + lhi)) = make_uint4(t0[0], t1v[0], t0[1], t1v[1]);
                if constexpr (OUT_R1) {      // (ck[j].x = the four residual bytes of the lane's channel quad j)
                    const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                    *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned char *>(out_c) + obase + 8 * (2 * m + lhi)) = make_uint2(u0[0], u0[1]);
                } else {
                    const auto u0 = __builtin_amdgcn_permlane32_swap(ck[0].x, ck[1].x, false, false);
                    const auto u1 = __builtin_amdgcn_permlane32_swap(ck[0].y, ck[1].y, false, false);
                    *reinterpret_cast<uint4 *>(out_c + obase + 8 * (2 * m + lhi)) = make_uint4(u0[0], u1[0], u0[1], u1[1]);
                }
            }
            { const unsigned int wb = sfd2_wave_max_bits(sfd2_h2_max(mx_out)); smax_out = wb > smax_out ? wb : smax_out; }
        };
        // four copies of the row's code: a loop would carry the residual registers around its back edge through copies, and the
        // compiler waits for the loads in front of those
        row(0, rqa, rca); R23_CYC(6)
        row(1, rqb, rcb); R23_CYC(7)
        row(2, rqa, rca); R23_CYC(8)
        row(3, rqb, rcb);
        R23_CYC(9)
        ++tcount;
#undef R23_RES
        if (!has_next) break;
        tile = next; oy0 = noy0; ox0 = nox0;
        // a full wait at every tile's top.  Until round 4 this was vmcnt(4) = "the last row's four stores are all that is in flight", which leans on
        // stores retiring in order with the filter-fragment loads issued just before them; the full wait measures the same (77.0 / 76.8 / 78.0 ->
        // 77.4 / 76.4 / 78.7 us per block) and leans on nothing
        drain = true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (copies of chunks nobody will read)
    sfd2_range_commit(range_t2, smax_t2);
    sfd2_range_commit(range_out, smax_out);
    R23_WALL(1)
#undef R23_COPIES
#undef R23_SETUP
}

// t1: ResBlock.conv1's output, plain fp16 [H][W][256]; res / res_c: the block's input (hi + corr planes); out / out_c: its output
void launch_rb23_c(hipStream_t st, const half_t *t1, int H, int W, const half_t *w2h, const half_t *w2l, const float *sc2,
                   const float *sh2, const half_t *w3h, const half_t *w3l, const float *sc3, const float *sh3,
                   const half_t *res, const half_t *res_c, half_t *out, half_t *out_c, const half_t *zero_page,
                   unsigned int *range_t2, unsigned int *range_out, int r1 /* bit 0: res_c is residual-only, bit 1: out_c is written residual-only */,
                   const half_t *w3l8, int sbyte3 /* r1 & 1: conv3's filter residuals as e4m3 ([8][4][64][32 B]) and its corr scale byte */)
{
    constexpr size_t lds = 0;                               // (static LDS: 150 KB)
    static bool attr_done = false;
    static int slots = 256;
    if (!attr_done) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            slots = cus;
        attr_done = true;
    }
    const int tiles_x = (W + R23_TW - 1) / R23_TW, tiles_y = (H + R23_TH - 1) / R23_TH;
    const int n_tiles = tiles_x * tiles_y;
    if (n_tiles == 0) return;
    const int grid = n_tiles < sfd2_slots(slots) ? n_tiles : sfd2_slots(slots);
    const int sa3 = (sbyte3 & 255) * 0x01010101;
    if ((r1 & 1) && !w3l8) abort();
#define R23_GO(a_, b_) hipLaunchKernelGGL((rb23_c_kernel<a_, b_>), dim3(grid), dim3(R23_NT), lds, st, t1, H, W, w2h, w2l, sc2, sh2, w3h, (a_) ? w3l8 : w3l, sc3, sh3, res, res_c, \
                                          out, out_c, tiles_x, n_tiles, range_t2, range_out, sa3)
    // (units in, residual bytes out is not instantiated: it spills 261 registers; the caller converts conv3b's output as well)
    if ((r1 & 3) == 3) R23_GO(true, true); else if (r1 & 1) R23_GO(true, false); else if (r1 & 2) abort(); else R23_GO(false, false);
#undef R23_GO
    (void)zero_page;
#ifdef SFD2_RB23_TRACE
    {
        static int dumps = 0;
        if (H >= 250 && ++dumps == 100) {
            (void)hipStreamSynchronize(st);
            static unsigned long long hc[2][6][12], hw[1024][2];
            (void)hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_r23_cyc), sizeof(hc));
            (void)hipMemcpyFromSymbol(hw, HIP_SYMBOL(g_r23_wall), sizeof(hw));
            unsigned long long t0 = ~0ull, e0 = ~0ull, e1 = 0; double es = 0;
            for (int b = 0; b < grid; ++b) t0 = hw[b][0] < t0 ? hw[b][0] : t0;
            for (int b = 0; b < grid; ++b) { const unsigned long long e = hw[b][1] - t0; e0 = e < e0 ? e : e0; e1 = e > e1 ? e : e1; es += (double)e; }
            fprintf(stderr, "rb23 trace %dx%d: %d tiles on %d blocks; exit (10 ns) min %llu mean %.0f max %llu\n", H, W, n_tiles, grid, e0, es / grid, e1);
            fprintf(stderr, "  columns: chunk 0 | 1 | 2 | 3 | residual + copies + barrier | row 0 | 1 | 2 | 3 | (next tile's wait)   cycles\n");
            for (int w = 0; w < 2; ++w)
                for (int t = 0; t < 4; ++t) {
                    fprintf(stderr, "  wave %d tile %d:", w * 7, t);
                    for (int k = 1; k < 10; ++k) fprintf(stderr, " %6lld", (long long)(hc[w][t][k] - hc[w][t][k - 1]));
                    fprintf(stderr, " [chunk 1: own wait %lld, then barrier %lld]", (long long)(hc[w][t][11] - hc[w][t][10]), (long long)(hc[w][t][1] - hc[w][t][11]));
                    fprintf(stderr, " | %6lld\n", (long long)(hc[w][t + 1][0] - hc[w][t][9]));
                }
        }
    }
#endif
}
