// Single-GEMM mutual nearest neighbour for the top-1 matcher modes.  Own translation unit because it is compiled with
// -fno-honor-nans (sfd2_amd/build.py): the maxima below run on bit-packed floats, and without that flag hipcc puts a
// NaN-canonicalising v_max_f32 x, x in front of every fmaxf operand it cannot prove quiet (4 VALU per element instead
// of 2).  Nothing in this file can produce a NaN: similarities of finite fp16 operands, MQ_NEG, and id bits or-ed into
// the low mantissa.
#include "sfd2_internal.h"
#include <math.h>

#define NT 256
#define KD 128
#define TA2 64        // candidate rows per LDS stage
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// The two directions of the mutual check are the row and the column maxima of ONE similarity matrix, so the top-1
// modes (NNM / ONN, it_loc nnm: hloc/matchers/nearest_neighbor.py:38-57, it_loc/matcher.py:122-130) need one GEMM, not
// two.  Orientation here: QUERIES are the MFMA rows (A operand, a wave keeps its 64 queries in registers), CANDIDATES
// the columns (B operand, streamed through LDS).  In the 32x32 C layout a lane owns 16 query rows of ONE candidate
// column, so per 32 x 32 sub-tile:
//   forward (best candidate per query): 16 element-wise running maxima per lane, kept across the whole sweep -- of the raw
//       accumulators in the mutual modes (the index comes from the reverse direction, see FWD_IDS below), else with the
//       candidate TILE id packed into the 7 low mantissa bits (127 - tile: the lower tile wins among equal values).  The
//       32-lane reduction happens once per sweep, through an LDS transposition, not per tile.
//   reverse (best query per candidate): a 16 -> 1 in-lane maximum with the row id packed into 5 bits, the two query
//       tiles and the two half-waves merged with two more id bits, and the block's four waves (two more) folded in a
//       per-block LDS table with ds_max_f32; the table leaves the CU once per sweep as a per-strip partial
//       [n0 / 256][n1] that match_mutual_reduce folds (13 MB per 50 x 4096^2 instead of 52 MB of per-wave partials).
// 5.6 VALU instructions per MFMA in the mutual modes (7.5 with forward ids) instead of a second GEMM.  Packing perturbs a
// similarity by <= 2^-15 relative (8 id bits, reverse; 2^-16 forward), below the fp16-operand error (1.5e-4); values equal
// after truncation -- exact ties included -- go to the lower index, as torch's argmax.
#ifdef SFD2_MQ_NO_TSWZ   /* the A side of the A/B: the forward transposition as before */
#define MQ_TSWZ(q_) 0
#else
#define MQ_TSWZ(q_) (((q_) >> 1) & 7)
#endif
#define MQ_NEG (-0x1p100f)    // "no value": finite with an all-zero mantissa, so or-ing id bits can only make it MORE negative
#define MQ_TILE_BITS 7
#define MQ_MAX_CHUNK (32 << MQ_TILE_BITS)   // candidates per split: the tile id must fit MQ_TILE_BITS

// FWD_IDS: the forward direction keeps the candidate (tile) id of its maximum -- the modes without a mutual check need
// matches0 for EVERY query.  With the mutual check a pair (i, j) is kept iff sim[i][j] is the maximum of row i AND of
// column j: the reverse direction's packed query id names i for every j, and the row maximum's VALUE decides
// (match_mutual_claim_kernel) -- the forward maxima are then plain v_max3_f32 over the raw accumulators, 0.5 instead of 1.5
// VALU per value.
template <bool FWD_IDS>
__global__ __launch_bounds__(NT, 2)
void match_mutual_kernel(const MatchJob2 *__restrict__ jobs, int splits, int qblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][TA2][256 B] (at the end [4][64][32] floats), then
                                                                           // the reverse table: MQ_MAX_CHUNK floats
    // XCD-aware work order.  Blocks are dealt round-robin to the 8 XCDs (each with its own L2); with the natural order the
    // query strips that share one candidate chunk (x fastest) land on all eight of them and the chunk is fetched into eight
    // L2s: 424 MB of HBM-side reads per 50 x 4096^2 against 52 MB of database sets (profiles/r02_match_pmc.txt).  Here every
    // XCD gets one contiguous run of the (pair, split, strip) list, so the strips of a chunk run back to back on one L2.
    int swz;
    {
        const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int bx = swz % qblocks, by = (swz / qblocks) % splits, bz = swz / (qblocks * splits);
    const MatchJob2 job = jobs[bz];
    const int n1 = job.n1, n0 = job.n0;          // candidates, queries
    const int i_base = bx * 256;
    if (i_base >= n0) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lcol = lane & 31, lhi = lane >> 5;

    int chunk = (n1 + splits - 1) / splits;
    chunk = (chunk + 31) & ~31;
    const int ja0 = by * chunk;
    int ja1 = ja0 + chunk;
    if (ja1 > n1) ja1 = n1;

    const int q0 = i_base + wave * 64;
    const bool wave_active = q0 < n0;                 // wave-uniform
    const bool partial_q = q0 + 64 > n0;              // some query row of this wave is padding
    h8_t aq[2][8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = q0 + t * 32 + lcol;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            h8_t z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (half_t)0.0f;
            aq[t][ks] = z;
            if (qi < n0) aq[t][ks] = *reinterpret_cast<const h8_t *>(job.q_hi + (size_t)qi * KD + ks * 16 + lhi * 8);
        }
    }
    float rm[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) rm[t][r] = MQ_NEG;
    f32x16_t zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    // reverse id = 255 - (query index within the block's 256): bits 7:6 = 3 - wave, 5 = 1 - t, 4:3 = 3 - (r >> 2), 2 = 1 - lhi,
    // 1:0 = 3 - (r & 3) (row of a 32 x 32 tile = 8 (r >> 2) + 4 lhi + (r & 3)).  The larger code wins the max, so among values
    // equal after truncation -- exact ties included: duplicated descriptors -- the LOWER query index wins, as torch's argmax
    const unsigned int wb = (3u - (unsigned)wave) << 6;
    const unsigned int cb[2] = {wb | 0x20u | ((1u - (unsigned)lhi) << 2), wb | ((1u - (unsigned)lhi) << 2)};
    // the block's reverse table: one packed maximum per candidate of this split
    float *rtab = reinterpret_cast<float *>(smem + 2 * TA2 * 256);
    for (int i = tid; i < ja1 - ja0; i += NT) rtab[i] = MQ_NEG;          // (published by the barrier behind the first stage's copies)
    // (issued by inline asm: for a compiler-visible LDS atomic hipcc first drains the next stage's direct-to-LDS copies,
    // s_waitcnt vmcnt(0), which it cannot prove disjoint from the table)
    const unsigned rt_lane = (unsigned)(size_t)(const lds_void_t *)rtab + (unsigned)lcol * 4u;
    // the mask lives in a VGPR so that (x & keep) | code is ONE v_and_or_b32 (VOP3 on gfx950 takes no literal, and the
    // tile code is already the one scalar operand)
    unsigned int keep = ~((1u << MQ_TILE_BITS) - 1u);
    asm volatile("" : "+v"(keep));

    // candidate fragments of one 32-candidate tile: LDS row SUB_ * 32 + lcol of buffer BUF_ (both constants: immediates)
    const unsigned char *brow = smem + lcol * 256;
    const int bsw = lcol & 15;
#define MQ_BFRAG(BUF_, SUB_, KS_) \
    (*reinterpret_cast<const h8_t *>(brow + ((BUF_)*TA2 + (SUB_)*32) * 256 + ((((KS_)*2 + lhi) ^ bsw) << 4)))
    // edge tiles only (wave-uniform, rare): padding candidate columns (zero rows) and padding query rows must never
    // be a maximum.  The empty asm keeps the block a branch: if-converted it costs 64 selects on every tile.
#define MQ_TILE_MASK(C0_, C1_, TILE_)                                                                    \
    if (ja0 + (TILE_)*32 + 32 > ja1 || partial_q) {                                                      \
        int qlim_ = n0 - q0 - 4 * lhi;      /* opaque: the 32 row comparisons stay in here (hoisted they hold 64 SGPRs) */ \
        asm volatile("" : "+v"(qlim_)::"memory");                                                        \
        const bool pad_col_ = ja0 + (TILE_)*32 + lcol >= ja1;                                            \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
            const int qr_ = (r & 3) + 8 * (r >> 2);                                                      \
            if (pad_col_ || qr_ >= qlim_) C0_[r] = MQ_NEG;                                               \
            if (pad_col_ || qr_ + 32 >= qlim_) C1_[r] = MQ_NEG;                                          \
        }                                                                                                \
    }
    // reverse direction of a finished tile: per-candidate maximum over the lane's 16 + 16 query rows (row id packed), the
    // two query tiles and the two half-waves merged, one ds_max_f32 into the block's table (the upper half-wave repeats
    // the lower one's: idempotent, no exec-mask branch; padding columns hold MQ_NEG and are not written out).
#define MQ_TILE_REV(C0_, C1_, TILE_)                                                                     \
    {                                                                                                    \
        float m0_ = MQ_NEG, m1_ = MQ_NEG;                                                                \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                 \
            m0_ = fmaxf(m0_, __uint_as_float((__float_as_uint(C0_[r]) & 0xFFFFFFE0u) | (unsigned)(((3 - (r >> 2)) << 3) | (3 - (r & 3))))); \
            m1_ = fmaxf(m1_, __uint_as_float((__float_as_uint(C1_[r]) & 0xFFFFFFE0u) | (unsigned)(((3 - (r >> 2)) << 3) | (3 - (r & 3))))); \
        }                                                                                                \
        const float k0_ = __uint_as_float((__float_as_uint(m0_) & 0xFFFFFF1Bu) | cb[0]);                 \
        const float k1_ = __uint_as_float((__float_as_uint(m1_) & 0xFFFFFF1Bu) | cb[1]);                 \
        const unsigned int kb_ = __float_as_uint(fmaxf(k0_, k1_));                                       \
        const auto sw2_ = __builtin_amdgcn_permlane32_swap(kb_, kb_, false, false);                      \
        const float kk_ = fmaxf(__uint_as_float(sw2_[0]), __uint_as_float(sw2_[1]));                     \
        asm volatile("ds_max_f32 %0, %1" ::"v"(rt_lane + (unsigned)(TILE_)*128u), "v"(kk_) : "memory");          \
    }
    // one stage = 64 candidates = two tiles: four accumulator chains (the matrix pipe measured 1.11 PFLOP/s with four
    // independent chains per wave against 0.90 with two: tools/probe/mfma_valu.hip), then both tiles' epilogue.  Forward:
    // FWD_IDS: element-wise running maxima with the tile id packed into the 7 low mantissa bits, else of the raw values --
    // either way ONE v_max3_f32 per query row takes both tiles.
// A wave raises its priority for the MFMA burst of a stage: two waves of a SIMD that start their bursts together otherwise share the matrix
// pipe AND then the vector ALU (MFMA time + VALU time per stage pair, as measured); with the priority the first one through keeps the pipe,
// the other follows, and from then on one wave's epilogue runs under the other's burst.  (-DSFD2_MQ_NO_PRIO: without)
#ifdef SFD2_MQ_NO_PRIO
#define MQ_PRIO(p_)
#else
#define MQ_PRIO(p_) __builtin_amdgcn_s_setprio(p_);
#endif
#define MQ_STAGE(S_, BUF_)                                                                               \
    {                                                                                                    \
        f32x16_t a0_, a1_, b0_, b1_;                                                                     \
        h8_t fa_ = MQ_BFRAG(BUF_, 0, 0), fb_ = MQ_BFRAG(BUF_, 1, 0), na_ = fa_, nb_ = fb_;               \
        MQ_PRIO(1)                                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                               \
            /* the fragments of K step ks + 1 are requested in front of the MFMAs of step ks */         \
            if (ks + 1 < 8) { na_ = MQ_BFRAG(BUF_, 0, ks + 1); nb_ = MQ_BFRAG(BUF_, 1, ks + 1); }        \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            a0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0][ks], fa_, ks == 0 ? zero16 : a0_, 0, 0, 0); \
            a1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[1][ks], fa_, ks == 0 ? zero16 : a1_, 0, 0, 0); \
            b0_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0][ks], fb_, ks == 0 ? zero16 : b0_, 0, 0, 0); \
            b1_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[1][ks], fb_, ks == 0 ? zero16 : b1_, 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            fa_ = na_; fb_ = nb_;                                                                        \
        }                                                                                                \
        MQ_PRIO(0)                                                                                       \
        const int ta_ = (S_)*2, tb_ = (S_)*2 + 1;                                                        \
        MQ_TILE_MASK(a0_, a1_, ta_)                                                                      \
        MQ_TILE_MASK(b0_, b1_, tb_)                                                                      \
        if (FWD_IDS) {                                                                                   \
            const unsigned int ca_ = (unsigned)((1 << MQ_TILE_BITS) - 1 - ta_), cb_ = ca_ - 1u;          \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                rm[0][r] = fmaxf(fmaxf(rm[0][r], __uint_as_float((__float_as_uint(a0_[r]) & keep) | ca_)), \
                                 __uint_as_float((__float_as_uint(b0_[r]) & keep) | cb_));               \
                rm[1][r] = fmaxf(fmaxf(rm[1][r], __uint_as_float((__float_as_uint(a1_[r]) & keep) | ca_)), \
                                 __uint_as_float((__float_as_uint(b1_[r]) & keep) | cb_));               \
            }                                                                                            \
        } else {                                                                                         \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                             \
                rm[0][r] = fmaxf(fmaxf(rm[0][r], a0_[r]), b0_[r]);                                       \
                rm[1][r] = fmaxf(fmaxf(rm[1][r], a1_[r]), b1_[r]);                                       \
            }                                                                                            \
        }                                                                                                \
        MQ_TILE_REV(a0_, a1_, ta_)                                                                       \
        MQ_TILE_REV(b0_, b1_, tb_)                                                                       \
    }

    if (ja0 < ja1) {
        const int nst = (ja1 - ja0 + TA2 - 1) / TA2;
        // staging copies: buffer_load ... lds, 16 B per lane; the per-lane byte offset of (row, 16-byte slot) advances by one
        // stage per stage, rows beyond ja1 fall outside the descriptor and read zeros (no zero page, no selects)
        const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t *>(job.d_hi), 0, ja1 * (KD * 2), 0x00020000);
        int voff[TA2 / 16];
#pragma unroll
        for (int c = 0; c < TA2 / 16; ++c) {
            const int row = (wave * (TA2 / 16) + c) * 4 + (lane >> 4);
            voff[c] = (ja0 + row) * (KD * 2) + (((lane & 15) ^ (row & 15)) << 4);
        }
#define ISSUE_B(buf_)                                                                                    \
    _Pragma("unroll") for (int c = 0; c < TA2 / 16; ++c) {                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, (lds_void_t *)(smem + (buf_)*TA2 * 256 + (wave * (TA2 / 16) + c) * 1024), \
                                                 16, voff[c], 0, 0, 0);                                  \
        voff[c] += TA2 * (KD * 2);                                                                       \
    }
        ISSUE_B(0)
        SFD2_BARRIER_DRAIN();
        // stage by stage, two per loop trip so that the buffer index is a constant (fragment addresses = per-lane base +
        // immediate).  Measured alternatives (profiles/r02_match_pmc.txt): the MFMAs of tile k + 1 software-pipelined
        // with the epilogue of tile k (+2 %, register pressure), the same pinned with sched_barrier (+3 %), 32 queries per
        // wave at 4 waves per SIMD (+12 %), 128-candidate stages (+23 %, spills), one wave per SIMD (+48 %); late round 2:
        // two wave groups one barrier apart as in conv3x3_pp (512 queries per block, MFMA section / epilogue section per tile,
        // three candidate stages in flight): bit-identical, 269 -> 322-341 us
        // round 3: the eight candidate-fragment LDS offsets hoisted into registers (the compiler already folds them into
        // ds_read immediates): 268.1/275.2/268.2 -> 267.8/262.9/269.8 us, within noise, not kept
        for (int s = 0; s < nst; s += 2) {
            if (s + 1 < nst) { ISSUE_B(1) }
            if (wave_active) MQ_STAGE(s, 0)
            SFD2_BARRIER_DRAIN();
            if (s + 1 < nst) {
                if (s + 2 < nst) { ISSUE_B(0) }
                if (wave_active) MQ_STAGE(s + 1, 1)
                SFD2_BARRIER_DRAIN();
            }
        }
#undef ISSUE_B
        __syncthreads();
        // the block's reverse partial: row bx of [n0 / 256][n1]
        float *rk = job.rkeys + (size_t)bx * n1 + ja0;
        for (int i = tid; i < ja1 - ja0; i += NT) rk[i] = rtab[i];
    }
#undef MQ_STAGE
#undef MQ_BFRAG
#undef MQ_TILE_MASK
#undef MQ_TILE_REV
    // ---- forward: reduce the 32 column classes of every query row through LDS (staging buffers are free now)
    float *T = reinterpret_cast<float *>(smem) + wave * (64 * 32);
    if (wave_active) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // float4 c of query row q sits at c ^ ((q >> 1) & 7): the sixteen rows one ds_read_b128 cycle serves below ({0-3, 12-15, 20-27}, ...:
                // MI355X_MICROARCH.md) are 128 B apart and would share two 16-byte slots (eight-way conflicts on all eight reads)
                const int q = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                T[(t * 32 + q) * 32 + (lcol ^ (MQ_TSWZ(q) << 2))] = rm[t][r];
            }
    }
    __syncthreads();
    if (wave_active && q0 + lane < n0) {
        float best = MQ_NEG;
        int col = 0;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const float4 v = *reinterpret_cast<const float4 *>(T + lane * 32 + ((c4 ^ MQ_TSWZ(lane)) << 2));
            if (v.x > best) { best = v.x; col = c4 * 4 + 0; }      // strict '>' in column order: lowest candidate among equals
            if (v.y > best) { best = v.y; col = c4 * 4 + 1; }
            if (v.z > best) { best = v.z; col = c4 * 4 + 2; }
            if (v.w > best) { best = v.w; col = c4 * 4 + 3; }
        }
        const unsigned int bits = __float_as_uint(best);
        const size_t o = (size_t)by * n0 + q0 + lane;
        const bool any = best > MQ_NEG;
        if (FWD_IDS) {
            const int tile = (1 << MQ_TILE_BITS) - 1 - (int)(bits & ((1u << MQ_TILE_BITS) - 1u));
            job.part_v1[o] = any ? __uint_as_float(bits & ~((1u << MQ_TILE_BITS) - 1u)) : -INFINITY;
            job.part_i1[o] = any ? ja0 + tile * 32 + col : 0;
        } else {
            job.part_v1[o] = any ? best : -INFINITY;           // the raw row maximum of this split
        }
    }
}

int match_mutual_strip(void) { return 256; }   // queries per reverse-partial strip (match_mutual_reduce decodes accordingly)

void launch_match_mutual_gemm(hipStream_t st, const MatchJob2 *jobs_dev, int npairs, int max_n0, int splits, int fwd_ids)
{
    static bool attr = false;
    const size_t lds = (size_t)2 * TA2 * 256 + (size_t)MQ_MAX_CHUNK * sizeof(float);
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_mutual_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(match_mutual_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int qblocks = (max_n0 + 255) / 256;
    if (fwd_ids) hipLaunchKernelGGL(match_mutual_kernel<true>, dim3(qblocks * splits * npairs), dim3(NT), lds, st, jobs_dev, splits, qblocks);
    else hipLaunchKernelGGL(match_mutual_kernel<false>, dim3(qblocks * splits * npairs), dim3(NT), lds, st, jobs_dev, splits, qblocks);
}

int match_mutual_max_chunk(void) { return MQ_MAX_CHUNK; }
