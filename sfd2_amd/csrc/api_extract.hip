// libsfd2hip: extraction entry points (sfd2_det, sfd2_extract, pyramids, spp variants, stage entry points).
#include "sfd2_ctx.h"

static int release_image_slot(sfd2_ctx *c);
// u8: 0 = fp32 [3][H][W], 1 = uint8 [H][W][3], 2 = uint8 [H][W][4] (SFD2_FLAG_IMG_U8_X: unpacked to three bytes on the device)
static int stage_image(sfd2_ctx *c, const void *x, int on_device, int H, int W, const float **dev, int u8 = 0)
{
    if (u8 == 2) {
        const float *raw = nullptr;
        if (on_device) raw = static_cast<const float *>(x);
        else if (stage_image(c, x, 0, H, W, &raw, 3)) return -1;          // (3: four bytes per pixel, staged as they are)
        HIPCHECK(c->img_u8_packed.ensure((size_t)3 * H * W + 16));
        launch_unpack_rgbx(c->stream, reinterpret_cast<const unsigned char *>(raw), c->img_u8_packed.as<unsigned char>(), (size_t)H * W);
        if (release_image_slot(c)) return -1;                             // the staged copy has been read once the unpack kernel is done
        *dev = c->img_u8_packed.as<float>();
        return 0;
    }
    if (on_device) { *dev = static_cast<const float *>(x); return 0; }
    const size_t bytes = u8 == 3 ? (size_t)4 * H * W : (size_t)3 * H * W * (u8 ? 1 : sizeof(float));
    const int slot = c->img_slot = (c->img_slot + 1) % SFD2_IMG_SLOTS;
    HIPCHECK(c->img2[slot].ensure(bytes));
    // the slot's previous reader (the network SFD2_IMG_SLOTS host images ago) must be done before the copy overwrites it
    HIPCHECK(hipStreamWaitEvent(c->copy_stream, c->ev_img_free[slot], 0));
    HIPCHECK(hipMemcpyAsync(c->img2[slot].p, x, bytes, hipMemcpyHostToDevice, c->copy_stream));
    HIPCHECK(hipEventRecord(c->ev_copied[slot], c->copy_stream));
    HIPCHECK(hipStreamWaitEvent(c->stream, c->ev_copied[slot], 0));
    c->img_slot_used = slot;
    *dev = c->img2[slot].as<float>();
    return 0;
}

// after the network that read a staged host image has been enqueued: its slot may be refilled once that work is done
static int release_image_slot(sfd2_ctx *c)
{
    if (c->img_slot_used >= 0) {
        HIPCHECK(hipEventRecord(c->ev_img_free[c->img_slot_used], c->stream));
        c->img_slot_used = -1;
    }
    return 0;
}

// Entry of an extraction that may fall back to SFD2_PREC_F16X3: remember where its profile steps start (the repeat replaces them)
// and, when the call is synchronous, move what earlier asynchronous calls left in the range words out of the way.
static int extract_begin(sfd2_ctx *c, int flags)
{
    c->prof_step_entry = c->prof_step;
    if (!(flags & SFD2_FLAG_ASYNC) && range_fold_before_sync_extract(c)) return -1;
    return 0;
}

int copy_out(sfd2_ctx *c, void *dst, const void *src_dev, size_t bytes, int dst_on_device)
{
    if (!dst || bytes == 0) return 0;
    HIPCHECK(hipMemcpyAsync(dst, src_dev, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, c->stream));
    return 0;
}

extern "C" int sfd2_det(sfd2_ctx *c, const float *x, int x_on_device, int H, int W, int flags, float *score,
                        float *stability, float *desc, int out_on_device, int *hs, int *ws, int *hc, int *wc)
{
    if (!c || !x) return fail("sfd2_det: null argument");
    if (!c->weights_loaded) return fail("sfd2_det: weights not loaded");
    if (stability && !c->has_sta) return fail("sfd2_det: stability requested but the loaded state_dict has no ConvSta");
    HIPCHECK(hipSetDevice(c->device));
    set_path(c, true);   // det is the parity entry point: every activation stays readable unless "fuse_det" is set
    if (ensure_workspace(c, H, W)) return -1;
    const float *img = nullptr;
    if (stage_image(c, x, x_on_device, H, W, &img)) return -1;
    prof_step_begin(c);
    if (run_network(c, img, (flags & SFD2_FLAG_IMG_NORMALISED) ? 0 : 1)) return -1;
    if (release_image_slot(c)) return -1;
    prof_step_end(c);
    const int HS = 8 * c->H8, WS = 8 * c->W8;
    if (hs) *hs = HS;
    if (ws) *ws = WS;
    if (hc) *hc = c->H4;
    if (wc) *wc = c->W4;
    if (copy_out(c, score, c->score.p, (size_t)HS * WS * sizeof(float), out_on_device)) return -1;
    if (stability) {
        HIPCHECK(c->stab.ensure((size_t)H * W * sizeof(float)));
        launch_heatmap(c->stream, c->score.as<float>(), HS, WS, c->sta.as<float>(), c->H4, c->W4, H, W, nullptr,
                       c->stab.as<float>());
        if (copy_out(c, stability, c->stab.p, (size_t)H * W * sizeof(float), out_on_device)) return -1;
    }
    if (desc) {
        const size_t n = (size_t)c->H4 * c->W4;
        HIPCHECK(c->desc_nchw.ensure(n * 128 * sizeof(float)));
        launch_desc_normalise_nchw(c->stream, c->draw.as<float>(), (int)n, c->desc_nchw.as<float>());
        if (copy_out(c, desc, c->desc_nchw.p, n * 128 * sizeof(float), out_on_device)) return -1;
    }
    HIPCHECK(hipGetLastError());
    if (!(flags & SFD2_FLAG_ASYNC)) HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// NMS + selection on c->heat; results in c->kpts / c->kscores, count in counters[1]
static int run_selection(sfd2_ctx *c, const float *heat_dev, int H, int W, float conf_th, int radius, int border,
                         int top_k, float *nms_dense, float *kpts_dev = nullptr, float *scores_dev = nullptr,
                         int Hb = 0, int Wb = 0)
{
    if (Hb <= 0) Hb = H;
    if (Wb <= 0) Wb = W;
    if (radius < 0 || radius > 4) return fail("nms radius must be in [0,4] (reference uses 4)");
    const int sel_cap = top_k > 0 ? std::min(top_k, c->cand_cap) : c->cand_cap;
    c->last_sel_cap = sel_cap;
    HIPCHECK(c->sel.ensure((size_t)sel_cap * 8));
    HIPCHECK(c->sorted.ensure((size_t)sel_cap * 8));
    HIPCHECK(c->kpts.ensure((size_t)sel_cap * 2 * sizeof(float)));
    HIPCHECK(c->kscores.ensure((size_t)sel_cap * sizeof(float)));
    if (!c->counters_clean) HIPCHECK(hipMemsetAsync(c->counters.p, 0, SFD2_COUNTER_BYTES, c->stream));
    c->counters_clean = false;
    bool threshold_done = false;
    {
        ProfScope ps(c, "nms_select", "nms_select_kernel", 0.0, (double)H * W * 4);
        threshold_done = launch_nms_select(c->stream, heat_dev, H, W, radius, conf_th, border, Hb, Wb, nms_dense,
                                           c->cand.as<unsigned long long>(), c->cand_cap, c->counters.as<unsigned int>(), 1, top_k);
    }
    {
        ProfScope ps(c, "topk_sort", "hist_select+compact+rank_sort", 0.0, (double)sel_cap * 24);
        c->kpts_cur = kpts_dev ? kpts_dev : c->kpts.as<float>();      // written in place when the caller's buffers are
        c->kscores_cur = scores_dev ? scores_dev : c->kscores.as<float>();   // device resident: no staging copies
        launch_topk_sort(c->stream, threshold_done, c->cand.as<unsigned long long>(), c->cand_cap, top_k,
                         c->sel.as<unsigned long long>(), c->sorted.as<unsigned long long>(), sel_cap,
                         c->counters.as<unsigned int>(), c->bnd.as<unsigned long long>(), W, c->kpts_cur, c->kscores_cur);
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

static int read_counts(sfd2_ctx *c, int64_t cap_out, int *n_out)
{
    unsigned int cnt[4] = {0, 0, 0, 0};
    HIPCHECK(hipMemcpyAsync(cnt, c->counters.p, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->tim.n_candidates = cnt[0];
    if (cnt[0] > (unsigned int)c->cand_cap)
        return fail("candidate buffer overflow: " + std::to_string(cnt[0]) + " NMS survivors > capacity " +
                    std::to_string(c->cand_cap));
    int64_t n = cnt[1];
    if (n > c->last_sel_cap) n = c->last_sel_cap;
    if (cap_out >= 0 && n > cap_out) n = cap_out;
    if (n_out) *n_out = (int)n;
    return 0;
}

extern "C" int sfd2_extract(sfd2_ctx *c, const void *img, int img_on_device, int H, int W, float conf_th, int top_k,
                            int flags, float *kpts_xy, float *scores, float *desc, int out_on_device, int64_t cap_out,
                            int *n_out)
{
    if (!c || !img) return fail("sfd2_extract: null argument");
    if (!c->weights_loaded) return fail("sfd2_extract: weights not loaded");
    if (!(flags & SFD2_FLAG_NO_STABILITY) && !c->has_sta)
        return fail("sfd2_extract: the loaded state_dict has no ConvSta; pass SFD2_FLAG_NO_STABILITY (use_stability=False)");
    HIPCHECK(hipSetDevice(c->device));
    if (extract_begin(c, flags)) return -1;
    set_path(c, false);
    if (ensure_workspace(c, H, W)) return -1;
    const float *img_dev = nullptr;
    const int u8 = (flags & SFD2_FLAG_IMG_U8_HWC) ? 1 : 0;
    if (u8 && (flags & SFD2_FLAG_IMG_NORMALISED)) return fail("sfd2_extract: a uint8 image cannot be pre-normalised");
    if ((flags & SFD2_FLAG_IMG_BGR) && !u8) return fail("sfd2_extract: SFD2_FLAG_IMG_BGR needs SFD2_FLAG_IMG_U8_HWC");
    if ((flags & SFD2_FLAG_IMG_U8_X) && !u8) return fail("sfd2_extract: SFD2_FLAG_IMG_U8_X needs SFD2_FLAG_IMG_U8_HWC");
    if (stage_image(c, img, img_on_device, H, W, &img_dev, u8 ? ((flags & SFD2_FLAG_IMG_U8_X) ? 2 : 1) : 0)) return -1;
    HIPCHECK(hipEventRecord(c->ev[0], c->stream));
    prof_step_begin(c);
    const int in_mode = ((flags & SFD2_FLAG_IMG_NORMALISED) ? 0 : 1) | (u8 ? 2 : 0) | ((flags & SFD2_FLAG_IMG_BGR) ? 4 : 0);
    // H, W multiples of 8 (every BASELINE geometry): the score map needs no resize, so the detector soft-max and the
    // stability weighting run as ONE kernel that writes the heat map directly; the score map is never materialised
    const bool fuse_post = c->opt_fuse_post && H % 8 == 0 && W % 8 == 0;
    c->skip_head_now = fuse_post ? 1 : 0;
    const bool fuse_pb = fuse_post && c->opt_fuse_pb && c->fuse_now && !c->opt_branches && c->pb.cout_pad >= 96;
    c->skip_pb_now = fuse_pb ? 1 : 0;
    // Sparse descriptor head: convDb is 1x1 and only the bilinear corners of the selected key points are sampled, so on the
    // throughput path it runs after the selection, on 4 x K gathered pixels instead of the whole 1/4-resolution map (the
    // 61 MB fp32 descriptor map is never written; bit-identical descriptors).  Dense when more than a quarter of the map
    // would be gathered (top_k <= 0: every candidate).
    const int sel_bound = top_k > 0 ? top_k : c->cand_cap;
    const bool sparse_desc = c->opt_sparse_desc && c->fuse_now && desc && top_k > 0 && (size_t)16 * sel_bound <= (size_t)c->H4 * c->W4;
    c->skip_db_now = sparse_desc ? 1 : 0;
    // ... and convDa.3 (3x3) is needed at those corners only as well: 4 x K pixels instead of the whole map (sparse_da3_kernel).
    // Not with compensated head branches (option "comp_heads": convDa.0's output then has a corr plane this kernel does not read).
    const bool comp_heads_now = c->precision == SFD2_PREC_F16C && c->opt_comp_heads && c->opt_comp_rb;
    const bool sparse_da3 = sparse_desc && c->opt_sparse_da3 && !comp_heads_now && !c->opt_branches;
    // SFD2_PREC_F16X3: the same two steps in that mode's arithmetic (planes of convDa.0's output, three MFMA passes, fp32 results)
    const bool sparse_x3 = c->precision == SFD2_PREC_F16X3 && c->opt_sparse_desc && c->opt_sparse_da3 && c->opt_x3_pp && desc && top_k > 0 &&
                           (size_t)16 * sel_bound <= (size_t)c->H4 * c->W4;
    c->skip_da3_now = (sparse_da3 || sparse_x3) ? 1 : 0;
    const int net_rc = run_network(c, img_dev, in_mode);
    c->skip_head_now = 0;
    c->skip_db_now = 0;
    c->skip_da3_now = 0;
    c->skip_pb_now = 0;
    if (net_rc) return -1;
    if (release_image_slot(c)) return -1;
    HIPCHECK(hipEventRecord(c->ev[1], c->stream));
    const int HS = 8 * c->H8, WS = 8 * c->W8;
    if (fuse_pb) {
        ProfScope ps(c, "convPb+heads+heatmap", "pb_heads_heat_kernel", 2.0 * c->H8 * c->W8 * 65 * 256,
                     (double)c->H8 * c->W8 * 512 + (double)H * W * 4);
        launch_pb_heads_heat(c->stream, c->pa_cur, c->H8, c->W8, c->pb.w.as<half_t>(), c->pb.cout_pad, c->pb.scale.as<float>(),
                             c->pb.shift.as<float>(), (flags & SFD2_FLAG_NO_STABILITY) ? nullptr : c->sta.as<float>(), c->H4, c->W4,
                             H, W, c->heat.as<float>(), c->counters.as<unsigned int>(), SFD2_COUNTER_BYTES / 4);
        c->counters_clean = pb_heads_heat_clears(c->H8, c->W8, SFD2_COUNTER_BYTES / 4);
    } else if (fuse_post) {
        ProfScope ps(c, "heads+heatmap", "heads_heat_kernel", 0.0, (double)c->H8 * c->W8 * 65 * 4 + (double)H * W * 4);
        launch_heads_heat(c->stream, c->logits.as<float>(), 128, c->H8, c->W8,
                          (flags & SFD2_FLAG_NO_STABILITY) ? nullptr : c->sta.as<float>(), c->H4, c->W4, H, W, c->heat.as<float>());
    } else {
        ProfScope ps(c, "heatmap", "heatmap_kernel", 0.0, (double)H * W * 8);
        launch_heatmap(c->stream, c->score.as<float>(), HS, WS,
                       (flags & SFD2_FLAG_NO_STABILITY) ? nullptr : c->sta.as<float>(), c->H4, c->W4, H, W,
                       c->heat.as<float>(), nullptr);
    }
    const int sel_guess = top_k > 0 ? std::min(top_k, c->cand_cap) : c->cand_cap;
    const bool direct = out_on_device && kpts_xy && scores && cap_out >= sel_guess;
    if (run_selection(c, c->heat.as<float>(), H, W, conf_th, 4, 4, top_k, nullptr, direct ? kpts_xy : nullptr,
                      direct ? scores : nullptr)) return -1;
    const int sel_cap = c->last_sel_cap;
    int64_t ncopy = sel_cap;
    if (cap_out >= 0 && ncopy > cap_out) ncopy = cap_out;
    const bool store64 = desc && (flags & SFD2_FLAG_DESC_STORE64);
    if (store64 && cap_out < 1) return fail("sfd2_extract: SFD2_FLAG_DESC_STORE64 needs cap_out >= 1 (desc is double [128][cap_out])");
    float *desc_dst = nullptr;
    if (desc) {
        if (out_on_device && cap_out >= sel_cap && !store64) {
            desc_dst = desc;  // sample straight into the caller's buffer
        } else {
            HIPCHECK(c->kdesc.ensure((size_t)sel_cap * 128 * sizeof(float)));
            desc_dst = c->kdesc.as<float>();
        }
        if (sparse_x3 && !c->x3_desc16_now) {
            const size_t nin = (size_t)c->H4 * c->W4 * 256;
            const int rows32 = (sel_cap * 4 + 31) / 32;           // the compact pixels as a [rows32][32] image for the generic 1x1 kernel
            HIPCHECK(c->da3_sparse.ensure((size_t)rows32 * 32 * 256 * sizeof(float)));
            HIPCHECK(c->db_sparse.ensure((size_t)rows32 * 32 * 128 * sizeof(float)));
            ConvW &L3 = c->fda3;
            if (!L3.wx3p.p) {
                const size_t nfl = (size_t)9 * L3.cout_pad * L3.cin;
                HIPCHECK(L3.wx3p.ensure(nfl * 2 * sizeof(half_t)));
                launch_x3_split_planes(c->stream, L3.w.as<float>(), nfl, L3.wx3p.p, L3.wx3p.as<half_t>() + nfl);
            }
            if (!L3.wsl.p && L3.cin == 256) {                     // both planes in sparse_da3_kernel's fragment order (ConvW::wsl)
                const size_t nfl = (size_t)9 * L3.cout_pad * L3.cin;
                HIPCHECK(L3.wsl.ensure(nfl * 2 * sizeof(half_t)));
                launch_sparse_da3_repack(c->stream, L3.wx3p.as<half_t>(), L3.wsl.as<half_t>(), L3.cout_pad, L3.cin);
                launch_sparse_da3_repack(c->stream, L3.wx3p.as<half_t>() + nfl, L3.wsl.as<half_t>() + nfl, L3.cout_pad, L3.cin);
            }
            {
                ProfScope ps(c, "convDa.3", "sparse_da3_kernel<x3>", 2.0 * 4 * sel_cap * 256.0 * 256.0 * 9, (double)sel_cap * (16 * 1024 + 4 * 1024));
                launch_sparse_da3_x3(c->stream, c->x3_da0_planes.as<half_t>(), c->x3_da0_planes.as<half_t>() + nin, c->H4, c->W4, H, W,
                                     L3.wx3p.as<half_t>(), L3.wsl.as<half_t>(), L3.cout_pad, L3.scale.as<float>(), L3.shift.as<float>(), 0, c->kpts_cur,
                                     c->counters.as<unsigned int>() + 1, sel_cap, c->da3_sparse.as<float>(), c->zero_page.as<half_t>());
            }
            // convDb (1x1) on the compact [sel_cap x 4] "image" with the mode's generic kernel, then the sampler on its compact output
            if (convf(c, "convDb", c->fdb, c->da3_sparse, rows32, 32, c->db_sparse, rows32, 32, 0)) return -1;
            ProfScope ps(c, "sample_desc", "sample_desc_kernel", 0.0, (double)sel_cap * 128 * 4 * 5);
            launch_sample_desc(c->stream, c->db_sparse.as<float>(), c->H4, c->W4, H, W, c->kpts_cur, c->counters.as<unsigned int>() + 1,
                               sel_cap, desc_dst, 1);
        } else if (sparse_da3 || (sparse_x3 && c->x3_desc16_now)) {      // (option "x3_desc16": f16x3's key points, the fp16 sparse head on convDa.0's fp16 output)
            HIPCHECK(c->da3_sparse.ensure((size_t)sel_cap * 4 * 256 * sizeof(half_t)));
            {
                ProfScope ps(c, "convDa.3", "sparse_da3_kernel", 2.0 * 4 * sel_cap * 256.0 * 256.0 * 9, (double)sel_cap * (16 * 512 + 4 * 512) + 2.0 * 256 * 256 * 9);
                launch_sparse_da3(c->stream, c->da0_cur, c->H4, c->W4, H, W, c->da3.w.as<half_t>(), c->da3.wsl.as<half_t>(), c->da3.cout_pad, c->da3.scale.as<float>(),
                                  c->da3.shift.as<float>(), 0, c->kpts_cur, c->counters.as<unsigned int>() + 1, sel_cap,
                                  c->da3_sparse.as<half_t>(), c->zero_page.as<half_t>());
            }
            ProfScope ps(c, "desc_head", "desc_head_kernel", 2.0 * 4 * sel_cap * 128 * 256, (double)sel_cap * (4 * 512 + 512));
            launch_desc_head(c->stream, c->da3_sparse.as<half_t>(), c->H4, c->W4, H, W, c->db.w.as<half_t>(), c->db.cout_pad, c->db.scale.as<float>(),
                             c->db.shift.as<float>(), c->kpts_cur, c->counters.as<unsigned int>() + 1, sel_cap, desc_dst, 1);
        } else if (sparse_desc) {
            ProfScope ps(c, "desc_head", "desc_head_kernel", 2.0 * 4 * sel_cap * 128 * 256, (double)sel_cap * (4 * 512 + 512));
            launch_desc_head(c->stream, c->da_cur, c->H4, c->W4, H, W, c->db.w.as<half_t>(), c->db.cout_pad, c->db.scale.as<float>(),
                             c->db.shift.as<float>(), c->kpts_cur, c->counters.as<unsigned int>() + 1, sel_cap, desc_dst);
        } else {
            ProfScope ps(c, "sample_desc", "sample_desc_kernel", 0.0, (double)sel_cap * 128 * 4 * 5);
            launch_sample_desc(c->stream, c->draw.as<float>(), c->H4, c->W4, H, W, c->kpts_cur,
                               c->counters.as<unsigned int>() + 1, sel_cap, desc_dst);
        }
    }
    size_t desc_bytes_fixed = 0;          // store64: the copy has one size whatever the count is
    if (store64) {
        float *d64_dst = out_on_device ? desc : nullptr;
        if (!out_on_device) {
            HIPCHECK(c->kdesc64.ensure((size_t)128 * cap_out * sizeof(double)));
            d64_dst = c->kdesc64.as<float>();
        }
        ProfScope ps(c, "desc_store64", "desc_store64_kernel", 0.0, (double)sel_cap * 128 * 4 + (double)cap_out * 128 * 8);
        launch_desc_store64(c->stream, desc_dst, c->counters.as<unsigned int>() + 1, sel_cap, reinterpret_cast<double *>(d64_dst), (int)cap_out);
        desc_dst = d64_dst;
        desc_bytes_fixed = (size_t)128 * cap_out * sizeof(double);
    }
    prof_step_end(c);
    HIPCHECK(hipEventRecord(c->ev[2], c->stream));
    HIPCHECK(hipGetLastError());
    if (flags & SFD2_FLAG_ASYNC) {
        // the fixed-capacity arrays are copied, the count stays on the device (sfd2_extract_record_async / sfd2_extract_count).
        // Host output buffers (pinned, or the copies are not asynchronous) hold the result once the stream has been synchronised.
        if (!direct && copy_out(c, kpts_xy, c->kpts.p, (size_t)ncopy * 2 * sizeof(float), out_on_device)) return -1;
        if (!direct && copy_out(c, scores, c->kscores.p, (size_t)ncopy * sizeof(float), out_on_device)) return -1;
        if (desc && desc_dst != desc && copy_out(c, desc, desc_dst, desc_bytes_fixed ? desc_bytes_fixed : (size_t)ncopy * 128 * sizeof(float), out_on_device)) return -1;
        if (n_out) *n_out = -1;
        return 0;
    }
    int n = 0;
    if (read_counts(c, cap_out, &n)) return -1;
    {   // SFD2_PREC_F16C: a tensor reached the saturation of the compensated format -> this image again in the strict arithmetic
        const int fb = range_wants_fallback(c);
        if (fb < 0) return -1;
        if (fb) {
            FallbackScope scope(c);
            return sfd2_extract(c, img, img_on_device, H, W, conf_th, top_k, flags, kpts_xy, scores, desc, out_on_device, cap_out, n_out);
        }
    }
    if (!direct && copy_out(c, kpts_xy, c->kpts.p, (size_t)n * 2 * sizeof(float), out_on_device)) return -1;
    if (!direct && copy_out(c, scores, c->kscores.p, (size_t)n * sizeof(float), out_on_device)) return -1;
    if (desc && desc_dst != desc && copy_out(c, desc, desc_dst, desc_bytes_fixed ? desc_bytes_fixed : (size_t)n * 128 * sizeof(float), out_on_device)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[2]) == hipSuccess) c->tim.ms_total = ms;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->tim.ms_backbone = ms;
    if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) c->tim.ms_post = ms;
    if (n_out) *n_out = n;
    return 0;
}

// ImageDataset.__getitem__ after the decoder (extract_localization.py:168-186): astype(float32), cubic resize to
// new_h x new_w when they differ from H x W, HWC -> CHW, / 255.  out_chw_dev: device, [3][new_h][new_w] fp32.
extern "C" int sfd2_preprocess(sfd2_ctx *c, const unsigned char *img_hwc, int on_device, int H, int W, int flags, int new_h,
                               int new_w, float *out_chw_dev)
{
    if (!c || !img_hwc || !out_chw_dev) return fail("sfd2_preprocess: null argument");
    if (H < 1 || W < 1 || new_h < 1 || new_w < 1) return fail("sfd2_preprocess: bad size");
    HIPCHECK(hipSetDevice(c->device));
    const float *staged = nullptr;
    if (stage_image(c, img_hwc, on_device, H, W, &staged, (flags & SFD2_FLAG_IMG_U8_X) ? 2 : 1)) return -1;
    launch_ingest_u8(c->stream, reinterpret_cast<const unsigned char *>(staged), H, W, (flags & SFD2_FLAG_IMG_BGR) ? 1 : 0, new_h,
                     new_w, out_chw_dev);
    if (release_image_slot(c)) return -1;
    HIPCHECK(hipGetLastError());
    if (!(flags & SFD2_FLAG_ASYNC)) HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// Scale pyramid (nets/extractor.py:118-236,322-330): every level runs the single-scale pipeline on the bilinearly
// resized normalised image, keeps its own top_k (the union's top_k is a subset of the levels' top_k), and the levels
// are merged by score on the device.  Reference quirks kept: the border test uses the ORIGINAL W, H in level
// coordinates (:181-184); key points are mapped back with x * W / nw in fp32 (:211-212); descriptors are sampled at
// level coordinates (:199-208); top_k <= 0 returns the plain concatenation, no global sort (:322).
extern "C" int sfd2_extract_multiscale(sfd2_ctx *c, const void *img, int img_on_device, int H, int W, const double *scales,
                                       int n_scales, float conf_th, int top_k, int flags, float *kpts_xy, float *scores,
                                       float *desc, int out_on_device, int64_t cap_out, int *n_out)
{
    if (!c || !img || !scales) return fail("sfd2_extract_multiscale: null argument");
    if (n_scales < 1 || n_scales > 8) return fail("sfd2_extract_multiscale: 1..8 scales");
    if (!c->weights_loaded) return fail("sfd2_extract_multiscale: weights not loaded");
    if (!(flags & SFD2_FLAG_NO_STABILITY) && !c->has_sta)
        return fail("sfd2_extract_multiscale: the loaded state_dict has no ConvSta; pass SFD2_FLAG_NO_STABILITY");
    if (flags & SFD2_FLAG_ASYNC) return fail("sfd2_extract_multiscale: SFD2_FLAG_ASYNC is not supported");
    if (flags & SFD2_FLAG_DESC_STORE64) return fail("sfd2_extract_multiscale: SFD2_FLAG_DESC_STORE64 is not supported (single-scale sfd2_extract only)");
    if (!kpts_xy || !scores) return fail("sfd2_extract_multiscale: kpts_xy and scores are required");
    HIPCHECK(hipSetDevice(c->device));
    if (extract_begin(c, 0)) return -1;
    const int u8 = (flags & SFD2_FLAG_IMG_U8_HWC) ? 1 : 0;
    if (u8 && (flags & SFD2_FLAG_IMG_NORMALISED)) return fail("sfd2_extract_multiscale: a uint8 image cannot be pre-normalised");
    if ((flags & SFD2_FLAG_IMG_BGR) && !u8) return fail("sfd2_extract_multiscale: SFD2_FLAG_IMG_BGR needs SFD2_FLAG_IMG_U8_HWC");
    if (flags & SFD2_FLAG_IMG_U8_X) return fail("sfd2_extract_multiscale: SFD2_FLAG_IMG_U8_X is taken by sfd2_extract and sfd2_preprocess only");
    const int in_mode = ((flags & SFD2_FLAG_IMG_NORMALISED) ? 0 : 1) | (u8 ? 2 : 0) | ((flags & SFD2_FLAG_IMG_BGR) ? 4 : 0);
    int nh[8], nw[8], cap[8], off[8];
    int cap_total = 0;
    for (int l = 0; l < n_scales; ++l) {
        nh[l] = scales[l] == 1.0 ? H : (int)((double)H * scales[l]);    // int(H * s), :122-123
        nw[l] = scales[l] == 1.0 ? W : (int)((double)W * scales[l]);
        if (nh[l] < 8 || nw[l] < 8) return fail("sfd2_extract_multiscale: a pyramid level is smaller than 8x8");
        const size_t P1 = (size_t)nh[l] * nw[l];
        const size_t cc = std::min(std::max<size_t>(65536, P1 / 8), P1);   // ensure_workspace's candidate capacity
        cap[l] = top_k > 0 ? (int)std::min<size_t>((size_t)top_k, cc) : (int)cc;
        off[l] = cap_total;
        cap_total += cap[l];
    }
    const float *img_dev = nullptr;
    if (stage_image(c, img, img_on_device, H, W, &img_dev, u8)) return -1;
    HIPCHECK(c->ms_kp.ensure((size_t)cap_total * 2 * sizeof(float)));
    HIPCHECK(c->ms_sc.ensure((size_t)cap_total * sizeof(float)));
    if (desc) HIPCHECK(c->ms_de.ensure((size_t)cap_total * 128 * sizeof(float)));
    HIPCHECK(c->ms_keys.ensure((size_t)cap_total * 8));
    HIPCHECK(c->ms_sorted.ensure((size_t)cap_total * 8));
    HIPCHECK(c->ms_cnt.ensure(64));
    HIPCHECK(hipMemsetAsync(c->ms_cnt.p, 0, 64, c->stream));
    unsigned int *level_count = c->ms_cnt.as<unsigned int>();        // [0..7] level counts, [8..] merge counters
    unsigned int *ms_counters = level_count + 8;
    HIPCHECK(hipEventRecord(c->ev[0], c->stream));
    for (int l = 0; l < n_scales; ++l) {
        const float *lvl_img = img_dev;
        int mode = in_mode;
        if (nh[l] != H || nw[l] != W) {
            HIPCHECK(c->img_scaled.ensure((size_t)3 * nh[l] * nw[l] * sizeof(float)));
            launch_norm_resize(c->stream, img_dev, in_mode, H, W, nh[l], nw[l], c->img_scaled.as<float>());
            lvl_img = c->img_scaled.as<float>();
            mode = 0;
        }
        set_path(c, false);
        if (ensure_workspace(c, nh[l], nw[l])) return -1;
        if (run_network(c, lvl_img, mode)) return -1;
        launch_heatmap(c->stream, c->score.as<float>(), 8 * c->H8, 8 * c->W8,
                       (flags & SFD2_FLAG_NO_STABILITY) ? nullptr : c->sta.as<float>(), c->H4, c->W4, nh[l], nw[l],
                       c->heat.as<float>(), nullptr);
        if (run_selection(c, c->heat.as<float>(), nh[l], nw[l], conf_th, 4, 4, top_k, nullptr, nullptr, nullptr, H, W)) return -1;
        if (c->last_sel_cap != cap[l]) return fail("sfd2_extract_multiscale: internal capacity mismatch");
        if (desc)
            launch_sample_desc(c->stream, c->draw.as<float>(), c->H4, c->W4, nh[l], nw[l], c->kpts_cur,
                               c->counters.as<unsigned int>() + 1, cap[l], c->ms_de.as<float>() + (size_t)off[l] * 128);
        launch_ms_append(c->stream, c->kpts_cur, c->kscores_cur, c->counters.as<unsigned int>() + 1, cap[l], W, nw[l], H, nh[l],
                         c->ms_kp.as<float>() + (size_t)off[l] * 2, c->ms_sc.as<float>() + off[l], level_count + l);
        // candidate-buffer overflow of this level is checked after the merge (counters are reused by the next level)
        HIPCHECK(hipMemcpyAsync(c->ms_cand_seen + l, c->counters.p, 4, hipMemcpyDeviceToHost, c->stream));
        c->ms_cand_cap[l] = c->cand_cap;
    }
    if (release_image_slot(c)) return -1;     // every level has read the staged image
    const int64_t want = top_k > 0 ? std::min<int64_t>(top_k, cap_total) : cap_total;
    const int64_t n_max = cap_out >= 0 ? std::min<int64_t>(want, cap_out) : want;
    float *kp_dst = kpts_xy, *sc_dst = scores, *de_dst = desc;
    if (!out_on_device) {
        HIPCHECK(c->kpts.ensure((size_t)std::max<int64_t>(n_max, 1) * 2 * sizeof(float)));
        HIPCHECK(c->kscores.ensure((size_t)std::max<int64_t>(n_max, 1) * sizeof(float)));
        kp_dst = c->kpts.as<float>();
        sc_dst = c->kscores.as<float>();
        if (desc) {
            HIPCHECK(c->kdesc.ensure((size_t)std::max<int64_t>(n_max, 1) * 128 * sizeof(float)));
            de_dst = c->kdesc.as<float>();
        }
    }
    launch_ms_merge(c->stream, n_scales, off, level_count, c->ms_kp.as<float>(), c->ms_sc.as<float>(),
                    desc ? c->ms_de.as<float>() : nullptr, cap_total, top_k, c->ms_keys.as<unsigned long long>(),
                    c->ms_sorted.as<unsigned long long>(), ms_counters, (int)n_max, kp_dst, sc_dst, de_dst);
    HIPCHECK(hipEventRecord(c->ev[2], c->stream));
    HIPCHECK(hipGetLastError());
    unsigned int n_dev = 0;
    HIPCHECK(hipMemcpyAsync(&n_dev, ms_counters + 2, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int l = 0; l < n_scales; ++l)
        if (c->ms_cand_seen[l] > (unsigned int)c->ms_cand_cap[l])
            return fail("candidate buffer overflow at pyramid level " + std::to_string(l));
    {
        const int fb = range_wants_fallback(c);
        if (fb < 0) return -1;
        if (fb) {
            FallbackScope scope(c);
            return sfd2_extract_multiscale(c, img, img_on_device, H, W, scales, n_scales, conf_th, top_k, flags, kpts_xy, scores, desc,
                                           out_on_device, cap_out, n_out);
        }
    }
    const int64_t n = n_max > 0 ? (int64_t)n_dev : 0;
    if (!out_on_device) {
        if (copy_out(c, kpts_xy, kp_dst, (size_t)n * 2 * sizeof(float), 0)) return -1;
        if (copy_out(c, scores, sc_dst, (size_t)n * sizeof(float), 0)) return -1;
        if (desc && copy_out(c, desc, de_dst, (size_t)n * 128 * sizeof(float), 0)) return -1;
        HIPCHECK(hipStreamSynchronize(c->stream));
    }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[2]) == hipSuccess) c->tim.ms_total = ms;
    if (n_out) *n_out = (int)n;
    return 0;
}

extern "C" int sfd2_extract_count(sfd2_ctx *c, int *n_out)
{
    if (!c) return fail("sfd2_extract_count: null ctx");
    HIPCHECK(hipSetDevice(c->device));
    return read_counts(c, -1, n_out);
}

// greedy NMS to the fixed point; result (kept ? heat : 0) in c->g_kept
static int run_greedy_nms(sfd2_ctx *c, const float *heat_dev, int H, int W, float conf_th, int dist)
{
    const int n = H * W;
    HIPCHECK(c->g_keys.ensure((size_t)n * 8));
    HIPCHECK(c->g_state0.ensure(n));
    HIPCHECK(c->g_state1.ensure(n));
    HIPCHECK(c->g_kept.ensure((size_t)n * sizeof(float)));
    HIPCHECK(c->counters.ensure(SFD2_COUNTER_BYTES));
    launch_greedy_init(c->stream, heat_dev, n, conf_th, c->g_keys.as<unsigned long long>(), c->g_state0.as<unsigned char>());
    unsigned char *sa = c->g_state0.as<unsigned char>(), *sb = c->g_state1.as<unsigned char>();
    unsigned int *und = c->counters.as<unsigned int>() + 8;   // scratch word, outside the selection counters
    const int max_iter = 1 << 20;                             // every sweep decides at least the best undecided candidate
    for (int it = 0; it < max_iter;) {
        unsigned int left = 0;
        for (int b = 0; b < 8 && it < max_iter; ++b, ++it) {  // 8 sweeps per host round trip
            HIPCHECK(hipMemsetAsync(und, 0, 4, c->stream));
            launch_greedy_iter(c->stream, c->g_keys.as<unsigned long long>(), sa, sb, H, W, dist, und);
            std::swap(sa, sb);
        }
        HIPCHECK(hipMemcpyAsync(&left, und, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHECK(hipStreamSynchronize(c->stream));
        if (left == 0) break;
        if (it >= max_iter) return fail("greedy NMS did not converge");
    }
    launch_greedy_final(c->stream, heat_dev, sa, n, c->g_kept.as<float>());
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sfd2_nms_fast(sfd2_ctx *c, const float *heat, int H, int W, float conf_th, int dist, float *kept_out)
{
    if (!c || !heat || !kept_out) return fail("sfd2_nms_fast: null argument");
    if (dist < 0 || dist > 16) return fail("sfd2_nms_fast: dist out of range");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(c->heat.ensure((size_t)H * W * sizeof(float)));
    HIPCHECK(hipMemcpyAsync(c->heat.p, heat, (size_t)H * W * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (run_greedy_nms(c, c->heat.as<float>(), H, W, conf_th, dist)) return -1;
    if (copy_out(c, kept_out, c->g_kept.p, (size_t)H * W * sizeof(float), 0)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_extract_spp(sfd2_ctx *c, const float *x, int x_on_device, int H, int W, float conf_th, int flags,
                                float *kpts_xy, float *scores, float *desc, int64_t cap_out, int *n_out,
                                float *heat_out, float *desc_full_out)
{
    if (!c || !x) return fail("sfd2_extract_spp: null argument");
    if (!c->weights_loaded) return fail("sfd2_extract_spp: weights not loaded");
    if (!(flags & SFD2_FLAG_NO_STABILITY) && !c->has_sta)
        return fail("sfd2_extract_spp: the loaded state_dict has no ConvSta; pass SFD2_FLAG_NO_STABILITY");
    HIPCHECK(hipSetDevice(c->device));
    if (extract_begin(c, 0)) return -1;
    set_path(c, false);
    if (ensure_workspace(c, H, W)) return -1;
    const float *img_dev = nullptr;
    if (stage_image(c, x, x_on_device, H, W, &img_dev)) return -1;
    prof_step_begin(c);
    if (run_network(c, img_dev, 0)) return -1;   // the caller normalised the image (extract.py:280-287)
    if (release_image_slot(c)) return -1;
    const int HS = 8 * c->H8, WS = 8 * c->W8;
    launch_heatmap(c->stream, c->score.as<float>(), HS, WS, (flags & SFD2_FLAG_NO_STABILITY) ? nullptr : c->sta.as<float>(),
                   c->H4, c->W4, H, W, c->heat.as<float>(), nullptr);
    if (run_greedy_nms(c, c->heat.as<float>(), H, W, conf_th, 4)) return -1;
    // kept map -> threshold (> 0: every kept score is >= conf_th > 0), border 4, sort; no top-K (extract.py:236-244)
    if (run_selection(c, c->g_kept.as<float>(), H, W, 0.0f, 0, 4, 0, nullptr)) return -1;
    prof_step_end(c);
    int n = 0;
    if (read_counts(c, cap_out, &n)) return -1;
    {
        const int fb = range_wants_fallback(c);
        if (fb < 0) return -1;
        if (fb) {
            FallbackScope scope(c);
            return sfd2_extract_spp(c, x, x_on_device, H, W, conf_th, flags, kpts_xy, scores, desc, cap_out, n_out, heat_out, desc_full_out);
        }
    }
    if (copy_out(c, kpts_xy, c->kpts.p, (size_t)n * 2 * sizeof(float), 0)) return -1;
    if (copy_out(c, scores, c->kscores.p, (size_t)n * sizeof(float), 0)) return -1;
    if (desc && n > 0) {
        HIPCHECK(c->kdesc.ensure((size_t)n * 128 * sizeof(float)));
        launch_sample_desc(c->stream, c->draw.as<float>(), c->H4, c->W4, H, W, c->kpts.as<float>(), nullptr, n, c->kdesc.as<float>());
        if (copy_out(c, desc, c->kdesc.p, (size_t)n * 128 * sizeof(float), 0)) return -1;
    }
    if (copy_out(c, heat_out, c->heat.p, (size_t)H * W * sizeof(float), 0)) return -1;
    if (desc_full_out) {
        const size_t np = (size_t)c->H4 * c->W4;
        HIPCHECK(c->desc_nchw.ensure(np * 128 * sizeof(float)));
        launch_desc_normalise_nchw(c->stream, c->draw.as<float>(), (int)np, c->desc_nchw.as<float>());
        if (copy_out(c, desc_full_out, c->desc_nchw.p, np * 128 * sizeof(float), 0)) return -1;
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (n_out) *n_out = n;
    return 0;
}

// extrat_spp_feats_multiscale (extract.py:87-201).  The level schedule (Python floats, round()) is the caller's; this
// runs it: level 0 is the given NORMALISED image, every further level is the bilinear resize (align_corners=False) of
// the PREVIOUS level (:186-188, whether or not that level was emitted), and for every emitted level: det -> score map
// resized to the level size, NO stability weighting (:115 ignores it), heat >= conf_th, greedy grid NMS radius 4,
// sort by confidence, 4-pixel border tested against the ORIGINAL W, H in level coordinates (:143-147), descriptors
// sampled at level coordinates and renormalised (:163-174).  Key points are returned in LEVEL coordinates with the
// per-level counts; the caller maps them back in float64 (x * W / nw, :176-177).  No top-K.
extern "C" int sfd2_extract_spp_levels(sfd2_ctx *c, const float *x, int x_on_device, int H, int W, int n_levels,
                                       const int32_t *nh, const int32_t *nw, const int32_t *emit, float conf_th, int flags,
                                       float *kpts_xy, float *scores, float *desc, int64_t cap_out, int32_t *level_count)
{
    if (!c || !x || !nh || !nw || !emit || !level_count) return fail("sfd2_extract_spp_levels: null argument");
    if (n_levels < 1 || n_levels > 64) return fail("sfd2_extract_spp_levels: 1..64 levels");
    if (!c->weights_loaded) return fail("sfd2_extract_spp_levels: weights not loaded");
    if (nh[0] != H || nw[0] != W) return fail("sfd2_extract_spp_levels: level 0 must be the image itself");
    (void)flags;
    HIPCHECK(hipSetDevice(c->device));
    if (extract_begin(c, 0)) return -1;
    const float *cur = nullptr;
    if (stage_image(c, x, x_on_device, H, W, &cur)) return -1;
    DevBuf lvl[2];          // ping-pong level images (released at the end: this entry point is not on the throughput path)
    int64_t total = 0;
    int rc = 0;
    for (int l = 0; l < n_levels && rc == 0; ++l) {
        level_count[l] = 0;
        if (nh[l] < 8 || nw[l] < 8) { rc = fail("sfd2_extract_spp_levels: a level is smaller than 8x8"); break; }
        if (l > 0) {
            DevBuf &dst = lvl[l & 1];
            if (dst.ensure((size_t)3 * nh[l] * nw[l] * sizeof(float)) != hipSuccess) { rc = fail("sfd2_extract_spp_levels: out of memory"); break; }
            launch_norm_resize(c->stream, cur, 0, nh[l - 1], nw[l - 1], nh[l], nw[l], dst.as<float>());
            cur = dst.as<float>();
        }
        if (!emit[l]) continue;
        set_path(c, false);
        if ((rc = ensure_workspace(c, nh[l], nw[l])) != 0) break;
        if ((rc = run_network(c, cur, 0)) != 0) break;
        launch_heatmap(c->stream, c->score.as<float>(), 8 * c->H8, 8 * c->W8, nullptr, c->H4, c->W4, nh[l], nw[l],
                       c->heat.as<float>(), nullptr);
        if ((rc = run_greedy_nms(c, c->heat.as<float>(), nh[l], nw[l], conf_th, 4)) != 0) break;
        if ((rc = run_selection(c, c->g_kept.as<float>(), nh[l], nw[l], 0.0f, 0, 4, 0, nullptr, nullptr, nullptr, H, W)) != 0) break;
        int n = 0;
        if ((rc = read_counts(c, -1, &n)) != 0) break;
        if (total + n > cap_out) { rc = fail("sfd2_extract_spp_levels: output capacity exceeded"); break; }
        if (n > 0) {
            if (copy_out(c, kpts_xy + 2 * total, c->kpts.p, (size_t)n * 2 * sizeof(float), 0) ||
                copy_out(c, scores + total, c->kscores.p, (size_t)n * sizeof(float), 0)) { rc = -1; break; }
            if (desc) {
                if (c->kdesc.ensure((size_t)n * 128 * sizeof(float)) != hipSuccess) { rc = fail("sfd2_extract_spp_levels: out of memory"); break; }
                launch_sample_desc(c->stream, c->draw.as<float>(), c->H4, c->W4, nh[l], nw[l], c->kpts.as<float>(), nullptr, n,
                                   c->kdesc.as<float>());
                if (copy_out(c, desc + 128 * total, c->kdesc.p, (size_t)n * 128 * sizeof(float), 0)) { rc = -1; break; }
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail("sfd2_extract_spp_levels: stream error"); break; }
        }
        level_count[l] = n;
        total += n;
    }
    (void)hipStreamSynchronize(c->stream);
    (void)release_image_slot(c);
    lvl[0].release();
    lvl[1].release();
    if (rc == 0 && hipGetLastError() != hipSuccess) rc = fail("sfd2_extract_spp_levels: kernel launch failed");
    if (rc == 0) {
        const int fb = range_wants_fallback(c);
        if (fb < 0) return -1;
        if (fb) {
            FallbackScope scope(c);
            return sfd2_extract_spp_levels(c, x, x_on_device, H, W, n_levels, nh, nw, emit, conf_th, flags, kpts_xy, scores, desc, cap_out, level_count);
        }
    }
    return rc;
}

// ------------------------------------------------------------------------------------------ stage entry points
static int heat_to_device(sfd2_ctx *c, const float *heat, int H, int W)
{
    const size_t bytes = (size_t)H * W * sizeof(float);
    HIPCHECK(c->heat.ensure(bytes));
    HIPCHECK(hipMemcpyAsync(c->heat.p, heat, bytes, hipMemcpyHostToDevice, c->stream));
    size_t cap = std::max<size_t>(65536, (size_t)H * W / 8);
    cap = std::min(cap, (size_t)H * W);
    c->cand_cap = (int)cap;
    HIPCHECK(c->cand.ensure(cap * 8));
    HIPCHECK(c->bnd.ensure(cap * 8));
    HIPCHECK(c->counters.ensure(SFD2_COUNTER_BYTES));
    return 0;
}

extern "C" int sfd2_simple_nms(sfd2_ctx *c, const float *heat, int H, int W, int radius, float *nms_out)
{
    if (!c || !heat || !nms_out) return fail("sfd2_simple_nms: null argument");
    if (radius < 0 || radius > 4) return fail("nms radius must be in [0,4]");
    HIPCHECK(hipSetDevice(c->device));
    if (heat_to_device(c, heat, H, W)) return -1;
    HIPCHECK(c->tmp_f32.ensure((size_t)H * W * sizeof(float)));
    HIPCHECK(hipMemsetAsync(c->counters.p, 0, SFD2_COUNTER_BYTES, c->stream));
    launch_nms_select(c->stream, c->heat.as<float>(), H, W, radius, 0.0f, 0, H, W, c->tmp_f32.as<float>(), nullptr, 0,
                      c->counters.as<unsigned int>());
    HIPCHECK(hipGetLastError());
    if (copy_out(c, nms_out, c->tmp_f32.p, (size_t)H * W * sizeof(float), 0)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_select_keypoints(sfd2_ctx *c, const float *heat, int H, int W, float conf_th, int radius, int border,
                                     int top_k, float *kpts_xy, float *scores, int64_t cap_out, int *n_out)
{
    if (!c || !heat) return fail("sfd2_select_keypoints: null argument");
    HIPCHECK(hipSetDevice(c->device));
    if (heat_to_device(c, heat, H, W)) return -1;
    if (run_selection(c, c->heat.as<float>(), H, W, conf_th, radius, border, top_k, nullptr)) return -1;
    int n = 0;
    if (read_counts(c, cap_out, &n)) return -1;
    if (copy_out(c, kpts_xy, c->kpts.p, (size_t)n * 2 * sizeof(float), 0)) return -1;
    if (copy_out(c, scores, c->kscores.p, (size_t)n * sizeof(float), 0)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (n_out) *n_out = n;
    return 0;
}

extern "C" int sfd2_sample_descriptors(sfd2_ctx *c, const float *desc_map, int hc, int wc, int nh, int nw,
                                       const float *kpts_xy, int n, float *desc_out)
{
    if (!c || !desc_map || !kpts_xy || !desc_out) return fail("sfd2_sample_descriptors: null argument");
    HIPCHECK(hipSetDevice(c->device));
    const size_t np = (size_t)hc * wc;
    HIPCHECK(c->tmp_f32.ensure(np * 128 * sizeof(float)));
    HIPCHECK(c->draw.ensure(np * 128 * sizeof(float)));
    HIPCHECK(c->kpts.ensure((size_t)std::max(n, 1) * 2 * sizeof(float)));
    HIPCHECK(c->kdesc.ensure((size_t)std::max(n, 1) * 128 * sizeof(float)));
    HIPCHECK(hipMemcpyAsync(c->tmp_f32.p, desc_map, np * 128 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHECK(hipMemcpyAsync(c->kpts.p, kpts_xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_nchw_f_to_nhwc_f(c->stream, c->tmp_f32.as<float>(), (int)np, 128, c->draw.as<float>());
    launch_sample_desc(c->stream, c->draw.as<float>(), hc, wc, nh, nw, c->kpts.as<float>(), nullptr, n, c->kdesc.as<float>());
    HIPCHECK(hipGetLastError());
    if (copy_out(c, desc_out, c->kdesc.p, (size_t)n * 128 * sizeof(float), 0)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_heatmap(sfd2_ctx *c, const float *score, int hs, int ws, const float *sta, int hc, int wc, int H,
                            int W, float *heat_out)
{
    if (!c || !score || !heat_out) return fail("sfd2_heatmap: null argument");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(c->score.ensure((size_t)hs * ws * sizeof(float)));
    HIPCHECK(c->heat.ensure((size_t)H * W * sizeof(float)));
    HIPCHECK(hipMemcpyAsync(c->score.p, score, (size_t)hs * ws * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (sta) {
        HIPCHECK(c->sta.ensure((size_t)3 * hc * wc * sizeof(float)));
        HIPCHECK(hipMemcpyAsync(c->sta.p, sta, (size_t)3 * hc * wc * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    launch_heatmap(c->stream, c->score.as<float>(), hs, ws, sta ? c->sta.as<float>() : nullptr, hc, wc, H, W,
                   c->heat.as<float>(), nullptr);
    HIPCHECK(hipGetLastError());
    if (copy_out(c, heat_out, c->heat.p, (size_t)H * W * sizeof(float), 0)) return -1;
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_debug_activation(sfd2_ctx *c, const char *name, float *out, int64_t cap, int *ch, int *h, int *w)
{
    if (!c || !name) return fail("sfd2_debug_activation: null argument");
    HIPCHECK(hipSetDevice(c->device));
    auto it = c->acts.find(name);
    if (it == c->acts.end()) return fail(std::string("unknown activation: ") + name);
    const ActInfo &a = it->second;
    if (!a.p || a.absent) return fail(std::string("activation not materialised on this path: ") + name);
    if (ch) *ch = a.c;
    if (h) *h = a.h;
    if (w) *w = a.w;
    if (!out) return 0;
    const size_t n = (size_t)a.c * a.h * a.w;
    if ((int64_t)n > cap) return fail("sfd2_debug_activation: output buffer too small");
    const int np = a.h * a.w;
    if (a.planar) {
        if (copy_out(c, out, a.p, n * sizeof(float), 0)) return -1;
    } else {
        HIPCHECK(c->tmp_f32.ensure(n * sizeof(float)));
        if (a.f32) launch_nhwc_f_to_nchw_f(c->stream, reinterpret_cast<const float *>(a.p), np, a.pitch, a.c, c->tmp_f32.as<float>());
        else if (a.pc) launch_nhwc_hc_to_nchw_f(c->stream, reinterpret_cast<const half_t *>(a.p), reinterpret_cast<const half_t *>(a.pc), np, a.pitch, a.c, c->tmp_f32.as<float>(), a.r1 ? 2 : (a.fmt6 ? 1 : 0));
        else launch_nhwc_h_to_nchw_f(c->stream, reinterpret_cast<const half_t *>(a.p), np, a.pitch, a.c, c->tmp_f32.as<float>());
        if (a.exp2) launch_scale_inplace(c->stream, c->tmp_f32.as<float>(), n, std::ldexp(1.0f, -a.exp2));
        if (copy_out(c, out, c->tmp_f32.p, n * sizeof(float), 0)) return -1;
    }
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}
