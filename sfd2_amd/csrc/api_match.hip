// libsfd2hip: matcher entry points.
#include "sfd2_ctx.h"

// ------------------------------------------------------------------------------------------ matcher
static size_t elt_size(int dtype) { return dtype == SFD2_DT_F64 ? 8 : (dtype == SFD2_DT_F16 ? 2 : 4); }

// Makes a device fp16 [n][128] view (hi, optional lo) of one descriptor set.
// n_src rows live in `src`; `rows` (host, n entries) selects and orders the ones that take part, or null = all n_src.
static int prep_set(sfd2_ctx *c, const void *src, int n_src, const int32_t *rows, int n, int dim, int dtype, int layout,
                    int on_device, int need_lo, DevBuf &stage, size_t &stage_off, half_t *hi_dst, half_t *lo_dst,
                    const half_t **hi, const half_t **lo, const int **rows_dev = nullptr)
{
    if (rows_dev) *rows_dev = nullptr;
    if (!rows && dtype == SFD2_DT_F16 && layout == SFD2_LAYOUT_ND && on_device && dim == 128 && !need_lo) {
        *hi = reinterpret_cast<const half_t *>(src);
        *lo = nullptr;
        return 0;
    }
    const void *dev_src = src;
    if (!on_device) {
        const size_t bytes = (size_t)n_src * dim * elt_size(dtype);
        void *dst = reinterpret_cast<char *>(stage.p) + stage_off;
        HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        dev_src = dst;
        stage_off += (bytes + 255) & ~(size_t)255;
    }
    const int *rd = nullptr;
    if (rows) {
        void *dst = reinterpret_cast<char *>(stage.p) + stage_off;
        HIPCHECK(hipMemcpyAsync(dst, rows, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        rd = reinterpret_cast<const int *>(dst);
        stage_off += ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;
        if (rows_dev) *rows_dev = rd;
    }
    launch_match_prep(c->stream, dev_src, n, n_src, rd, dim, dtype, layout, hi_dst, need_lo ? lo_dst : nullptr);
    *hi = hi_dst;
    *lo = need_lo ? lo_dst : nullptr;
    return 0;
}

// One descriptor set into the matcher's resident form (include/sfd2_hip.h): the conversion every matcher call does, done once.
extern "C" int sfd2_desc_pack(sfd2_ctx *c, const sfd2_desc_set *src, int dim, void *dst_f16_dev, int flags)
{
    if (!c || !src || !dst_f16_dev) return fail("sfd2_desc_pack: null argument");
    if (dim <= 0 || dim > 128) return fail("descriptor dimension must be in [1,128]");
    if (src->n < 0 || src->n_rows < 0) return fail("negative size");
    if (src->dtype != SFD2_DT_F32 && src->dtype != SFD2_DT_F64 && src->dtype != SFD2_DT_F16) return fail("sfd2_desc_pack: unknown dtype");
    if (src->layout != SFD2_LAYOUT_ND && src->layout != SFD2_LAYOUT_DN) return fail("sfd2_desc_pack: unknown layout");
    const int n = src->rows ? src->n_rows : src->n;
    if (n == 0) return 0;
    if (!src->data) return fail("sfd2_desc_pack: null descriptors");
    if (src->rows)
        for (int r = 0; r < src->n_rows; ++r)
            if (src->rows[r] < 0 || src->rows[r] >= src->n) return fail("sfd2_desc_pack: row index out of range");
    HIPCHECK(hipSetDevice(c->device));
    size_t stage_bytes = 256;
    if (!src->on_device) stage_bytes += (((size_t)src->n * dim * elt_size(src->dtype)) + 255) & ~(size_t)255;
    if (src->rows) stage_bytes += ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;
    HIPCHECK(c->m_stage.ensure(stage_bytes));
    size_t off = 0;
    void *dev_src = const_cast<void *>(src->data);
    if (!src->on_device) {
        const size_t bytes = (size_t)src->n * dim * elt_size(src->dtype);
        HIPCHECK(hipMemcpyAsync(c->m_stage.p, src->data, bytes, hipMemcpyHostToDevice, c->stream));
        dev_src = c->m_stage.p;
        off += (bytes + 255) & ~(size_t)255;
    }
    const int *rd = nullptr;
    if (src->rows) {
        void *dst = c->m_stage.as<char>() + off;
        HIPCHECK(hipMemcpyAsync(dst, src->rows, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        rd = reinterpret_cast<const int *>(dst);
    }
    // (always through the conversion kernel: a resident fp16 [n][128] source is copied, with dim < 128 zero-filled like every other)
    launch_match_prep(c->stream, dev_src, n, src->n, rd, dim, src->dtype, src->layout, reinterpret_cast<half_t *>(dst_f16_dev), nullptr);
    HIPCHECK(hipGetLastError());
    if (!(flags & SFD2_FLAG_ASYNC)) HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_match_batch(sfd2_ctx *c, const sfd2_desc_set *q, const sfd2_desc_set *db, int k, int dim,
                                const sfd2_match_conf *conf, int64_t *matches0, float *scores0, int out_on_device,
                                int flags)
{
    if (!c || !conf || !q || (k > 0 && !db)) return fail("sfd2_match_batch: null argument");
    if (dim <= 0 || dim > 128) return fail("descriptor dimension must be in [1,128]");
    const int n0 = q->n;
    if (n0 < 0 || k < 0) return fail("negative size");
    if (k == 0 || n0 == 0) return 0;
    if (!q->data) return fail("sfd2_match_batch: null query descriptors");
    HIPCHECK(hipSetDevice(c->device));
    const int need_lo = conf->sim_mode == SFD2_SIM_F16X2;
    const bool o16 = (flags & SFD2_FLAG_MATCH_OUT16) != 0;       // outputs as int16 / fp16 (converted by match_decide_kernel)
    const size_t msz = o16 ? sizeof(short) : sizeof(long long), ssz = o16 ? sizeof(half_t) : sizeof(float);
    int max_n1 = 0;
    size_t tot_n1 = 0, stage_bytes = 0;
    if (q->rows) return fail("sfd2_match_batch: row selection applies to database sets only");
    std::vector<int> eff_n1(k);
    for (int i = 0; i < k; ++i) {
        if (db[i].n < 0 || db[i].n_rows < 0) return fail("negative n1");
        if (db[i].n > 0 && !db[i].data) return fail("sfd2_match_batch: null database descriptors");
        if (db[i].rows)
            for (int r = 0; r < db[i].n_rows; ++r)
                if (db[i].rows[r] < 0 || db[i].rows[r] >= db[i].n) return fail("sfd2_match_batch: row index out of range");
        eff_n1[i] = db[i].rows ? db[i].n_rows : db[i].n;
        max_n1 = std::max(max_n1, eff_n1[i]);
        tot_n1 += (size_t)eff_n1[i];
        if (!db[i].on_device) stage_bytes += (((size_t)db[i].n * dim * elt_size(db[i].dtype)) + 255) & ~(size_t)255;
        if (db[i].rows) stage_bytes += ((size_t)db[i].n_rows * sizeof(int32_t) + 255) & ~(size_t)255;
    }
    if (!q->on_device) stage_bytes += (((size_t)n0 * dim * elt_size(q->dtype)) + 255) & ~(size_t)255;
    const int max_n = std::max(n0, max_n1);
    // splits of the candidate range: aim at >= 6 work items per resident block slot (256 CUs x 3 blocks)
    // so the last wave of blocks costs little, never finer than 32 candidates (measured: 1 -> 524 us,
    // 3-4 -> 470 us for 50 pairs of 4096 x 4096)
    const int blocks_per_job = (max_n + 255) / 256;
    int splits = (768 * 6 + 2 * k * blocks_per_job - 1) / (2 * k * blocks_per_job);
    splits = std::max(1, std::min(splits, 8));
    const int min_n = std::max(1, std::min(n0, max_n1 > 0 ? max_n1 : 1));
    splits = std::min(splits, std::max(1, (min_n + 31) / 32));
    if (const char *e = sfd2_env("SFD2_MATCH_SPLITS")) splits = std::max(1, std::min(16, atoi(e)));

    const int need_top2 = (conf->flavour == SFD2_MATCH_ITLOC_NNR) ||
                          (conf->flavour == SFD2_MATCH_HLOC && conf->ratio_threshold > 0.0f);
    // Top-1 modes (NNM / ONN / it_loc nnm) take both directions from ONE GEMM (match_mutual_kernel: element-wise running
    // maxima for the row direction, in-lane 16 -> 1 maxima for the column direction).  SFD2_MATCH_TWO_GEMM (experiment
    // builds) restores the two-GEMM kernel for A/B runs; modes that need the second-best value keep it.
    const bool single_gemm = !need_lo && !need_top2 && sfd2_env("SFD2_MATCH_TWO_GEMM") == nullptr;
    if (single_gemm) {
        // one GEMM per pair instead of two: twice the blocks per job for the same tail behaviour
        const int qblocks = (n0 + 255) / 256;
        // (512 resident blocks: two per CU at this kernel's 200 registers.  Measured for 50 x 4096^2, us: 3 splits 209.0, 4 207.9, 5 210.1,
        //  6 214.9, 8 220.2 -- six items per slot)
        splits = (512 * 6 + k * qblocks - 1) / (k * qblocks);
        splits = std::max(1, std::min(splits, 8));
        splits = std::min(splits, std::max(1, (std::max(1, max_n1) + 31) / 32));
        if (const char *e = sfd2_env("SFD2_MATCH_SPLITS")) splits = std::max(1, std::min(16, atoi(e)));
        splits = std::max(splits, (max_n1 + match_mutual_max_chunk() - 1) / match_mutual_max_chunk());   // tile id bits
    }
    const int nstrip = (n0 + match_mutual_strip() - 1) / match_mutual_strip();

    HIPCHECK(c->m_stage.ensure(std::max<size_t>(stage_bytes, 256)));
    HIPCHECK(c->m_hi0.ensure((size_t)n0 * 128 * 2));
    if (need_lo) HIPCHECK(c->m_lo0.ensure((size_t)n0 * 128 * 2));
    HIPCHECK(c->m_hi1.ensure(std::max<size_t>(tot_n1, 1) * 128 * 2));
    if (need_lo) HIPCHECK(c->m_lo1.ensure(std::max<size_t>(tot_n1, 1) * 128 * 2));
    // partials: per pair forward [splits][n0] and reverse [splits][n1], 2 float arrays + 1 int array
    const size_t per_pair_f = (size_t)splits * n0, tot_part = (size_t)k * per_pair_f + (size_t)splits * tot_n1;
    HIPCHECK(c->m_part_f.ensure(std::max<size_t>(tot_part, 1) * 2 * sizeof(float)));
    HIPCHECK(c->m_part_i.ensure(std::max<size_t>(tot_part, 1) * sizeof(int)));
    if (single_gemm) HIPCHECK(c->m_rkeys.ensure(std::max<size_t>((size_t)nstrip * tot_n1, 1) * sizeof(float)));
    HIPCHECK(c->m_red.ensure(((size_t)k * n0 + tot_n1 + 1) * 3 * sizeof(float)));
    // job and final descriptors share one device block (jobs first), filled by ONE copy from the pinned block
    HIPCHECK(c->m_jobs.ensure((size_t)2 * k * sizeof(MatchJob) + (size_t)k * sizeof(MatchFinal)));
    HIPCHECK(c->m_out_m.ensure((size_t)k * n0 * sizeof(long long)));
    HIPCHECK(c->m_out_s.ensure((size_t)k * n0 * sizeof(float)));

    HIPCHECK(hipEventRecord(c->ev[0], c->stream));
    prof_step_begin(c);
    size_t stage_off = 0;
    const half_t *q_hi = nullptr, *q_lo = nullptr;
    if (prep_set(c, q->data, n0, nullptr, n0, dim, q->dtype, q->layout, q->on_device, need_lo, c->m_stage, stage_off,
                 c->m_hi0.as<half_t>(), c->m_lo0.as<half_t>(), &q_hi, &q_lo)) return -1;
    // job descriptors live in pinned host memory owned by the context; the event makes sure the
    // previous call's async copies have consumed them before they are rewritten
    static_assert(sizeof(MatchJob2) <= 2 * sizeof(MatchJob), "descriptor buffer sizing");
    const size_t jobs_bytes = 2 * (size_t)k * sizeof(MatchJob), fins_bytes = (size_t)k * sizeof(MatchFinal);
    // hipGraph capture of the caller's stream (tools/graph_replay.py): no host-side event waits while capturing; the
    // captured copies read the pinned descriptors at replay time, so they must not be rewritten by eager calls meanwhile
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(c->stream, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    if (!capturing) HIPCHECK(hipEventSynchronize(c->ev_jobs));
    if (jobs_bytes + fins_bytes > c->pin_cap) {
        if (c->pin_jobs) (void)hipHostFree(c->pin_jobs);
        c->pin_jobs = nullptr;
        c->pin_cap = 0;
        HIPCHECK(hipHostMalloc(&c->pin_jobs, jobs_bytes + fins_bytes, hipHostMallocDefault));
        c->pin_cap = jobs_bytes + fins_bytes;
    }
    MatchJob *jobs = reinterpret_cast<MatchJob *>(c->pin_jobs);
    MatchJob2 *jobs2 = reinterpret_cast<MatchJob2 *>(c->pin_jobs);
    MatchFinal *fins = reinterpret_cast<MatchFinal *>(reinterpret_cast<char *>(c->pin_jobs) + jobs_bytes);
    float *pf = c->m_part_f.as<float>();
    int *pi = c->m_part_i.as<int>();
    float *red = c->m_red.as<float>();
    size_t off1 = 0, poff = 0, roff = 0;
    for (int i = 0; i < k; ++i) {
        const int n1 = eff_n1[i];
        const half_t *h = nullptr, *l = nullptr;
        const int *remap = nullptr;
        if (n1 > 0) {
            if (prep_set(c, db[i].data, db[i].n, db[i].rows, n1, dim, db[i].dtype, db[i].layout, db[i].on_device, need_lo,
                         c->m_stage, stage_off, c->m_hi1.as<half_t>() + off1 * 128,
                         need_lo ? c->m_lo1.as<half_t>() + off1 * 128 : nullptr, &h, &l, &remap))
                return -1;
        }
        MatchFinal &fn = fins[i];
        fn.remap = remap;
        if (single_gemm) {
            MatchJob2 &j2 = jobs2[i];
            j2.q_hi = q_hi; j2.d_hi = h; j2.n0 = n0; j2.n1 = n1;
            j2.part_v1 = pf + 2 * poff; j2.part_i1 = pi + poff;
            j2.rkeys = c->m_rkeys.as<float>() + (size_t)nstrip * off1;
            poff += (size_t)splits * n0 + (size_t)splits * n1;
            fn.f_v1 = fn.f_v2 = fn.r_v1 = fn.r_v2 = nullptr; fn.f_i1 = fn.r_i1 = nullptr;
        } else {
            MatchJob &f = jobs[2 * i], &r = jobs[2 * i + 1];
            // forward: keep queries (d0), reduce over d1
            f.a_hi = h; f.a_lo = l; f.b_hi = q_hi; f.b_lo = q_lo; f.na = n1; f.nb = n0;
            f.part_v1 = pf + 2 * poff; f.part_v2 = pf + 2 * poff + (size_t)splits * n0; f.part_i1 = pi + poff;
            poff += (size_t)splits * n0;
            // reverse: keep d1 rows, reduce over queries
            r.a_hi = q_hi; r.a_lo = q_lo; r.b_hi = h; r.b_lo = l; r.na = n0; r.nb = n1;
            r.part_v1 = pf + 2 * poff; r.part_v2 = pf + 2 * poff + (size_t)splits * n1; r.part_i1 = pi + poff;
            poff += (size_t)splits * n1;
            fn.f_v1 = f.part_v1; fn.f_v2 = f.part_v2; fn.f_i1 = f.part_i1;
            fn.r_v1 = r.part_v1; fn.r_v2 = r.part_v2; fn.r_i1 = r.part_i1;
        }
        off1 += (size_t)n1;
        fn.n0 = n0; fn.n1 = n1;
        fn.out16 = o16 ? 1 : 0; fn.pad_ = 0;
        const bool direct_out = out_on_device && matches0 && scores0;
        fn.matches0 = reinterpret_cast<long long *>((direct_out ? reinterpret_cast<char *>(matches0) : c->m_out_m.as<char>()) + (size_t)i * n0 * msz);
        fn.scores0 = reinterpret_cast<float *>((direct_out ? reinterpret_cast<char *>(scores0) : c->m_out_s.as<char>()) + (size_t)i * n0 * ssz);
        fn.red_f = red + 3 * roff; roff += (size_t)n0;
        fn.red_r = red + 3 * roff; roff += (size_t)n1;
    }
    HIPCHECK(hipMemcpyAsync(c->m_jobs.p, jobs, jobs_bytes + fins_bytes, hipMemcpyHostToDevice, c->stream));
    MatchFinal *fins_dev = reinterpret_cast<MatchFinal *>(static_cast<char *>(c->m_jobs.p) + jobs_bytes);
    if (!capturing) HIPCHECK(hipEventRecord(c->ev_jobs, c->stream));
    if (single_gemm) {
        {
            ProfScope ps(c, "match_mutual", "match_mutual_kernel", 2.0 * (double)n0 * (double)tot_n1 * 128.0,
                         2.0 * ((double)k * n0 + (double)tot_n1) * 128.0);
            launch_match_mutual(c->stream, c->m_jobs.as<MatchJob2>(), fins_dev, k, n0, max_n1, splits,
                                conf->flavour == SFD2_MATCH_HLOC && !conf->do_mutual_check);
        }
        {
            ProfScope ps(c, "match_finalize", "match_decide", 0.0, (double)tot_part * 12);
            launch_match_decide(c->stream, fins_dev, k, max_n, conf->flavour, conf->do_mutual_check,
                                conf->ratio_threshold, conf->distance_threshold);
        }
    } else {
        {
            // both directions: 2 GEMMs of n0 x n1 x 128 per pair (x3 products in the hi+lo mode)
            ProfScope ps(c, "match_top2", need_lo ? "match_top2_kernel<x2>" : "match_top2_kernel",
                         2.0 * 2.0 * (double)n0 * (double)tot_n1 * 128.0 * (need_lo ? 3.0 : 1.0),
                         2.0 * 2.0 * ((double)k * n0 + (double)tot_n1) * 128.0);
            launch_match_top2(c->stream, c->m_jobs.as<MatchJob>(), 2 * k, max_n, splits, need_lo, need_top2,
                              c->zero_page.as<half_t>());
        }
        {
            ProfScope ps(c, "match_finalize", "match_reduce+decide", 0.0, (double)tot_part * 12);
            launch_match_finalize(c->stream, fins_dev, k, max_n, splits, conf->flavour,
                                  conf->do_mutual_check, conf->ratio_threshold, conf->distance_threshold);
        }
    }
    prof_step_end(c);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(c->ev[3], c->stream));
    if (!(out_on_device && matches0 && scores0)) {
        if (copy_out(c, matches0, c->m_out_m.p, (size_t)k * n0 * msz, out_on_device)) return -1;
        if (copy_out(c, scores0, c->m_out_s.p, (size_t)k * n0 * ssz, out_on_device)) return -1;
    }
    if (!(flags & SFD2_FLAG_ASYNC)) {
        HIPCHECK(hipStreamSynchronize(c->stream));
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[3]) == hipSuccess) c->tim.ms_match = ms;
    }
    return 0;
}

// Segmented matcher: rows [seg0[s], seg0[s+1]) of d0 are matched against rows [seg1[s], seg1[s+1]) of d1 only -- the
// diagonal blocks of a block-masked similarity matrix, all segments in ONE launch of the matcher kernels (grid z =
// segment).  This is the same-label phase of the label-aware matcher (it_loc/matcher.py:248-264) once both descriptor
// sets are ordered by label.  matches0[i] = row of d1 (global index) or -1; scores0 as sfd2_match, within the segment.
extern "C" int sfd2_match_segments(sfd2_ctx *c, const void *d0, int n0, const void *d1, int n1, int dim, int dtype, int layout,
                                   int on_device, int n_seg, const int32_t *seg0, const int32_t *seg1,
                                   const sfd2_match_conf *conf, int64_t *matches0, float *scores0, int out_on_device)
{
    if (!c || !conf || !seg0 || !seg1 || !matches0 || !scores0) return fail("sfd2_match_segments: null argument");
    if (dim <= 0 || dim > 128) return fail("descriptor dimension must be in [1,128]");
    if (n0 < 0 || n1 < 0 || n_seg < 0) return fail("negative size");
    if (n0 == 0) return 0;
    if (!d0 || (n1 > 0 && !d1)) return fail("sfd2_match_segments: null descriptors");
    if (seg0[0] != 0 || seg1[0] != 0 || seg0[n_seg] != n0 || seg1[n_seg] != n1) return fail("sfd2_match_segments: segment offsets must span [0, n]");
    for (int i = 0; i < n_seg; ++i)
        if (seg0[i + 1] < seg0[i] || seg1[i + 1] < seg1[i]) return fail("sfd2_match_segments: segment offsets must ascend");
    HIPCHECK(hipSetDevice(c->device));
    const int need_lo = conf->sim_mode == SFD2_SIM_F16X2;
    const int need_top2 = (conf->flavour == SFD2_MATCH_ITLOC_NNR) || (conf->flavour == SFD2_MATCH_HLOC && conf->ratio_threshold > 0.0f);
    const bool single_gemm = !need_lo && !need_top2;
    std::vector<int> live;                     // segments with rows on both sides
    int max_a = 0, max_b = 0;
    for (int i = 0; i < n_seg; ++i)
        if (seg0[i + 1] > seg0[i] && seg1[i + 1] > seg1[i]) {
            live.push_back(i);
            max_a = std::max(max_a, seg0[i + 1] - seg0[i]);
            max_b = std::max(max_b, seg1[i + 1] - seg1[i]);
        }
    const int k = (int)live.size();
    // rows without a partner segment: no match
    HIPCHECK(c->m_out_m.ensure((size_t)n0 * sizeof(long long)));
    HIPCHECK(c->m_out_s.ensure((size_t)n0 * sizeof(float)));
    long long *out_m = out_on_device ? reinterpret_cast<long long *>(matches0) : c->m_out_m.as<long long>();
    float *out_s = out_on_device ? scores0 : c->m_out_s.as<float>();
    HIPCHECK(hipMemsetAsync(out_m, 0xFF, (size_t)n0 * sizeof(long long), c->stream));     // -1
    HIPCHECK(hipMemsetAsync(out_s, 0, (size_t)n0 * sizeof(float), c->stream));
    if (k > 0) {
        int splits = std::max(1, (max_b + match_mutual_max_chunk() - 1) / match_mutual_max_chunk());
        const int strip = match_mutual_strip();
        size_t stage_bytes = 256 + (((size_t)n1 * sizeof(int32_t)) + 255);
        if (!on_device) stage_bytes += ((((size_t)n0 + n1) * dim * elt_size(dtype)) + 511);
        HIPCHECK(c->m_stage.ensure(stage_bytes));
        HIPCHECK(c->m_hi0.ensure((size_t)n0 * 128 * 2));
        HIPCHECK(c->m_hi1.ensure((size_t)std::max(n1, 1) * 128 * 2));
        if (need_lo) { HIPCHECK(c->m_lo0.ensure((size_t)n0 * 128 * 2)); HIPCHECK(c->m_lo1.ensure((size_t)std::max(n1, 1) * 128 * 2)); }
        const size_t tot_part = (size_t)splits * ((size_t)n0 + n1);
        HIPCHECK(c->m_part_f.ensure(tot_part * 2 * sizeof(float)));
        HIPCHECK(c->m_part_i.ensure(tot_part * sizeof(int)));
        size_t rk = 0;
        for (int i : live) rk += (size_t)((seg0[i + 1] - seg0[i] + strip - 1) / strip) * (size_t)(seg1[i + 1] - seg1[i]);
        if (single_gemm) HIPCHECK(c->m_rkeys.ensure(std::max<size_t>(rk, 1) * sizeof(float)));
        HIPCHECK(c->m_red.ensure(((size_t)n0 + n1 + 1) * 3 * sizeof(float)));
        HIPCHECK(c->m_jobs.ensure((size_t)2 * k * sizeof(MatchJob) + (size_t)k * sizeof(MatchFinal)));
        prof_step_begin(c);
        size_t stage_off = 0;
        const half_t *h0 = nullptr, *l0 = nullptr, *h1 = nullptr, *l1 = nullptr;
        // forced conversion into the context's fp16 buffers (a device-resident fp16 set would otherwise be used in place,
        // which is fine too: the jobs only need row-offset pointers)
        if (prep_set(c, d0, n0, nullptr, n0, dim, dtype, layout, on_device, need_lo, c->m_stage, stage_off, c->m_hi0.as<half_t>(),
                     c->m_lo0.as<half_t>(), &h0, &l0)) return -1;
        if (prep_set(c, d1, n1, nullptr, n1, dim, dtype, layout, on_device, need_lo, c->m_stage, stage_off, c->m_hi1.as<half_t>(),
                     c->m_lo1.as<half_t>(), &h1, &l1)) return -1;
        // identity column map: remap + seg1[s] turns a segment-local match into the global row of d1
        std::vector<int32_t> iota((size_t)n1);
        for (int i = 0; i < n1; ++i) iota[i] = i;
        int32_t *iota_dev = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(c->m_stage.p) + ((stage_off + 255) & ~(size_t)255));
        HIPCHECK(hipMemcpyAsync(iota_dev, iota.data(), (size_t)n1 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));      // iota is a host temporary
        const size_t jobs_bytes = 2 * (size_t)k * sizeof(MatchJob), fins_bytes = (size_t)k * sizeof(MatchFinal);
        HIPCHECK(hipEventSynchronize(c->ev_jobs));
        if (jobs_bytes + fins_bytes > c->pin_cap) {
            if (c->pin_jobs) (void)hipHostFree(c->pin_jobs);
            c->pin_jobs = nullptr;
            c->pin_cap = 0;
            HIPCHECK(hipHostMalloc(&c->pin_jobs, jobs_bytes + fins_bytes, hipHostMallocDefault));
            c->pin_cap = jobs_bytes + fins_bytes;
        }
        MatchJob *jobs = reinterpret_cast<MatchJob *>(c->pin_jobs);
        MatchJob2 *jobs2 = reinterpret_cast<MatchJob2 *>(c->pin_jobs);
        MatchFinal *fins = reinterpret_cast<MatchFinal *>(reinterpret_cast<char *>(c->pin_jobs) + jobs_bytes);
        float *pf = c->m_part_f.as<float>();
        int *pi = c->m_part_i.as<int>();
        float *red = c->m_red.as<float>();
        size_t poff = 0, roff = 0, rkoff = 0;
        for (int j = 0; j < k; ++j) {
            const int sg = live[j], a0 = seg0[sg], na = seg0[sg + 1] - a0, b0 = seg1[sg], nb = seg1[sg + 1] - b0;
            MatchFinal &fn = fins[j];
            fn.out16 = 0; fn.pad_ = 0;
            fn.remap = iota_dev + b0;
            if (single_gemm) {
                MatchJob2 &j2 = jobs2[j];
                j2.q_hi = h0 + (size_t)a0 * 128; j2.d_hi = h1 + (size_t)b0 * 128; j2.n0 = na; j2.n1 = nb;
                j2.part_v1 = pf + 2 * poff; j2.part_i1 = pi + poff;
                j2.rkeys = c->m_rkeys.as<float>() + rkoff;
                rkoff += (size_t)((na + strip - 1) / strip) * nb;
                poff += (size_t)splits * na + (size_t)splits * nb;
                fn.f_v1 = fn.f_v2 = fn.r_v1 = fn.r_v2 = nullptr; fn.f_i1 = fn.r_i1 = nullptr;
            } else {
                MatchJob &f = jobs[2 * j], &r = jobs[2 * j + 1];
                f.a_hi = h1 + (size_t)b0 * 128; f.a_lo = l1 ? l1 + (size_t)b0 * 128 : nullptr;
                f.b_hi = h0 + (size_t)a0 * 128; f.b_lo = l0 ? l0 + (size_t)a0 * 128 : nullptr; f.na = nb; f.nb = na;
                f.part_v1 = pf + 2 * poff; f.part_v2 = pf + 2 * poff + (size_t)splits * na; f.part_i1 = pi + poff;
                poff += (size_t)splits * na;
                r.a_hi = f.b_hi; r.a_lo = f.b_lo; r.b_hi = f.a_hi; r.b_lo = f.a_lo; r.na = na; r.nb = nb;
                r.part_v1 = pf + 2 * poff; r.part_v2 = pf + 2 * poff + (size_t)splits * nb; r.part_i1 = pi + poff;
                poff += (size_t)splits * nb;
                fn.f_v1 = f.part_v1; fn.f_v2 = f.part_v2; fn.f_i1 = f.part_i1;
                fn.r_v1 = r.part_v1; fn.r_v2 = r.part_v2; fn.r_i1 = r.part_i1;
            }
            fn.n0 = na; fn.n1 = nb;
            fn.matches0 = out_m + a0;
            fn.scores0 = out_s + a0;
            fn.red_f = red + 3 * roff; roff += (size_t)na;
            fn.red_r = red + 3 * roff; roff += (size_t)nb;
        }
        HIPCHECK(hipMemcpyAsync(c->m_jobs.p, jobs, jobs_bytes + fins_bytes, hipMemcpyHostToDevice, c->stream));
        MatchFinal *fins_dev = reinterpret_cast<MatchFinal *>(static_cast<char *>(c->m_jobs.p) + jobs_bytes);
        HIPCHECK(hipEventRecord(c->ev_jobs, c->stream));
        const int max_n = std::max(max_a, max_b);
        if (single_gemm) {
            ProfScope ps(c, "match_segments", "match_mutual_kernel", 0.0, 0.0);
            launch_match_mutual(c->stream, c->m_jobs.as<MatchJob2>(), fins_dev, k, max_a, max_b, splits,
                                conf->flavour == SFD2_MATCH_HLOC && !conf->do_mutual_check);
            launch_match_decide(c->stream, fins_dev, k, max_n, conf->flavour, conf->do_mutual_check,
                                conf->ratio_threshold, conf->distance_threshold);
        } else {
            ProfScope ps(c, "match_segments", "match_top2_kernel", 0.0, 0.0);
            launch_match_top2(c->stream, c->m_jobs.as<MatchJob>(), 2 * k, max_n, splits, need_lo, need_top2, c->zero_page.as<half_t>());
            launch_match_finalize(c->stream, fins_dev, k, max_n, splits, conf->flavour, conf->do_mutual_check,
                                  conf->ratio_threshold, conf->distance_threshold);
        }
        prof_step_end(c);
        HIPCHECK(hipGetLastError());
    }
    if (!out_on_device) {
        if (copy_out(c, matches0, c->m_out_m.p, (size_t)n0 * sizeof(long long), 0)) return -1;
        if (copy_out(c, scores0, c->m_out_s.p, (size_t)n0 * sizeof(float), 0)) return -1;
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_match(sfd2_ctx *c, const void *d0, int n0, const void *d1, int n1, int dim, int dtype, int layout,
                          int on_device, const sfd2_match_conf *conf, int64_t *matches0, float *scores0, int out_on_device)
{
    const sfd2_desc_set q = {d0, n0, dtype, layout, on_device, nullptr, 0, 0};
    const sfd2_desc_set db = {d1, n1, dtype, layout, on_device, nullptr, 0, 0};
    return sfd2_match_batch(c, &q, &db, 1, dim, conf, matches0, scores0, out_on_device, 0);
}
