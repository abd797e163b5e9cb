// libsfd2hip: workspace, per-layer kernel dispatch and the network passes (ResSegNetV2.det up to the head outputs).
#include "sfd2_ctx.h"

// ------------------------------------------------------------------------------------------ workspace
static int down2(int n) { return (n - 1) / 2 + 1; }  // 3x3 stride 2 pad 1

void set_path(sfd2_ctx *c, bool parity_entry)
{
    const bool f16 = c->precision == SFD2_PREC_F16 || c->precision == SFD2_PREC_F16C;
    c->fuse_now = f16 && (parity_entry ? c->fuse_det : c->fuse);
    c->alias_now = c->fuse_now && !parity_entry && c->opt_alias;
    c->x3_fast_rb_now = c->precision == SFD2_PREC_F16X3 && !parity_entry && c->opt_x3_pp;
}

// Buffers are allocated for the path that is about to run only: the throughput path needs the 3-slot
// arena + head outputs, the layer-wise paths one buffer per activation, strict mode the fp32 set.
int ensure_workspace(sfd2_ctx *c, int H, int W)
{
    if (H < 8 || W < 8) return fail("image too small (need H, W >= 8)");
    if ((long long)H * W > (1ll << 30)) return fail("image too large");
    c->H = H; c->W = W;
    c->H2 = down2(H); c->W2 = down2(W);
    c->H4 = down2(c->H2); c->W4 = down2(c->W2);
    c->H8 = down2(c->H4); c->W8 = down2(c->W4);
    const size_t P1 = (size_t)H * W, P2 = (size_t)c->H2 * c->W2, P4 = (size_t)c->H4 * c->W4, P8 = (size_t)c->H8 * c->W8;
    const size_t hb = sizeof(half_t);
    const bool f32 = c->precision == SFD2_PREC_F32 || c->precision == SFD2_PREC_F16X3;   // they share the fp32 buffers
    const bool comp = c->precision == SFD2_PREC_F16C;
    const bool layers = !f32 && !c->alias_now;      // private fp16 buffer per activation
    // SFD2_PREC_F16C: every backbone activation is a hi plane followed by its corr plane (same geometry)
    const size_t bb = comp ? 2 * hb : hb;
    const bool fused_stem = c->fuse_now && !(comp && c->opt_generic_c);
    const bool fused_rb = c->fuse_now && (!comp || !c->opt_comp_rb);
    if (!f32 && !fused_stem) HIPCHECK(c->a1a.ensure(P1 * 64 * bb));
    if (layers) {
        HIPCHECK(c->a1b.ensure(P2 * 64 * bb));
        HIPCHECK(c->a2a.ensure(P2 * 128 * bb));
        HIPCHECK(c->a2b.ensure(P4 * 128 * bb));
        HIPCHECK(c->a3a.ensure(P4 * 256 * bb));
        HIPCHECK(c->a3b.ensure(P4 * 256 * bb));
        for (int b = 0; b < 3; ++b) {
            if (!fused_rb) {
                HIPCHECK(c->rt1[b].ensure(P4 * 256 * bb));
                HIPCHECK(c->rt2[b].ensure(P4 * 256 * bb));
            }
            HIPCHECK(c->ro[b].ensure(P4 * 256 * bb));
        }
        HIPCHECK(c->pa0_o.ensure(P8 * 256 * bb));   // (corr planes too with option "comp_heads")
        HIPCHECK(c->pa_o.ensure(P8 * 256 * hb));
        HIPCHECK(c->da0_o.ensure(P4 * 256 * bb));
        HIPCHECK(c->da_o.ensure(P4 * 256 * hb));
    }
    HIPCHECK(c->logits.ensure(P8 * 128 * sizeof(float)));
    HIPCHECK(c->draw.ensure(P4 * 128 * sizeof(float)));
    HIPCHECK(c->sta.ensure(P4 * 3 * sizeof(float)));
    HIPCHECK(c->score.ensure(P8 * 64 * sizeof(float)));
    HIPCHECK(c->heat.ensure(P1 * sizeof(float)));
    size_t cap = std::max<size_t>(65536, P1 / 8);
    cap = std::min(cap, P1);
    c->cand_cap = (int)cap;
    HIPCHECK(c->cand.ensure(cap * 8));
    HIPCHECK(c->bnd.ensure(cap * 8));
    HIPCHECK(c->counters.ensure(SFD2_COUNTER_BYTES));
    c->acts.clear();
    auto reg = [&](const char *nm, const void *ptr, int is_f32, int planar, int ch, int pitch, int h, int w) {
        ActInfo ai{ptr, is_f32, planar, ch, pitch, h, w};
        // backbone activations of SFD2_PREC_F16C: the corr plane follows the hi plane
        if (comp && !is_f32 && std::strncmp(nm, "convP", 5) != 0 && std::strncmp(nm, "convD", 5) != 0)
            ai.pc = reinterpret_cast<const half_t *>(ptr) + (size_t)pitch * h * w;
        if (!is_f32) {      // the fp16 family stores its tensors times 2^act_exp (sfd2_ctx.h): sfd2_debug_activation divides it out
            static const struct { const char *name; int group; } gmap[] = {
                {"conv1a", AE_CONV1A}, {"bn1b", AE_CONV1B}, {"conv2a", AE_CONV2A}, {"bn2b", AE_CONV2B}, {"conv3a", AE_CONV3A},
                {"bn3b", AE_TRUNK}, {"conv4.0", AE_TRUNK}, {"conv4.1", AE_TRUNK}, {"conv4.2", AE_TRUNK},
                {"conv4.0.bn1", AE_T1_0}, {"conv4.1.bn1", AE_T1_1}, {"conv4.2.bn1", AE_T1_2},
                {"conv4.0.bn2", AE_T2_0}, {"conv4.1.bn2", AE_T2_1}, {"conv4.2.bn2", AE_T2_2}, {"convPa.0", AE_PA0}, {"convDa.0", AE_DA0}};
            for (const auto &g : gmap)
                if (std::strcmp(nm, g.name) == 0) ai.exp2 = c->act_exp[g.group];
        }
        c->acts[nm] = ai;
    };
    static const char *n1[3] = {"conv4.0.bn1", "conv4.1.bn1", "conv4.2.bn1"};
    static const char *n2[3] = {"conv4.0.bn2", "conv4.1.bn2", "conv4.2.bn2"};
    static const char *n3[3] = {"conv4.0", "conv4.1", "conv4.2"};
    reg("convPb", c->logits.p, 1, 0, 65, 128, c->H8, c->W8);
    reg("convDb", c->draw.p, 1, 0, 128, 128, c->H4, c->W4);
    reg("ConvSta", c->sta.p, 1, 1, 3, 0, c->H4, c->W4);
    if (f32) {
        const size_t fb = sizeof(float);
        HIPCHECK(c->g1a.ensure(P1 * 64 * fb));
        HIPCHECK(c->g1b.ensure(P2 * 64 * fb));
        HIPCHECK(c->g2a.ensure(P2 * 128 * fb));
        HIPCHECK(c->g2b.ensure(P4 * 128 * fb));
        HIPCHECK(c->g3a.ensure(P4 * 256 * fb));
        HIPCHECK(c->g3b.ensure(P4 * 256 * fb));
        for (int b = 0; b < 3; ++b) {
            HIPCHECK(c->grt1[b].ensure(P4 * 256 * fb));
            HIPCHECK(c->grt2[b].ensure(P4 * 256 * fb));
            HIPCHECK(c->gro[b].ensure(P4 * 256 * fb));
        }
        HIPCHECK(c->gpa0_o.ensure(P8 * 256 * fb));
        HIPCHECK(c->gpa_o.ensure(P8 * 256 * fb));
        HIPCHECK(c->gda0_o.ensure(P4 * 256 * fb));
        HIPCHECK(c->gda_o.ensure(P4 * 256 * fb));
        reg("conv1a", c->g1a.p, 1, 0, 64, 64, H, W);
        reg("bn1b", c->g1b.p, 1, 0, 64, 64, c->H2, c->W2);
        reg("conv2a", c->g2a.p, 1, 0, 128, 128, c->H2, c->W2);
        reg("bn2b", c->g2b.p, 1, 0, 128, 128, c->H4, c->W4);
        reg("conv3a", c->g3a.p, 1, 0, 256, 256, c->H4, c->W4);
        reg("bn3b", c->g3b.p, 1, 0, 256, 256, c->H4, c->W4);
        for (int b = 0; b < 3; ++b) {
            reg(n1[b], c->grt1[b].p, 1, 0, 256, 256, c->H4, c->W4);
            reg(n2[b], c->grt2[b].p, 1, 0, 256, 256, c->H4, c->W4);
            reg(n3[b], c->gro[b].p, 1, 0, 256, 256, c->H4, c->W4);
        }
        reg("convPa.0", c->gpa0_o.p, 1, 0, 256, 256, c->H8, c->W8);
        reg("convPa", c->gpa_o.p, 1, 0, 256, 256, c->H8, c->W8);
        reg("convDa.0", c->gda0_o.p, 1, 0, 256, 256, c->H4, c->W4);
        reg("convDa", c->gda_o.p, 1, 0, 256, 256, c->H4, c->W4);
        if (c->x3_fast_rb_now) {
            // throughput path of SFD2_PREC_F16X3 (sfd2_extract): from the stem to convDa.0 the tensors exist as hi / lo' planes only; the fp32
            // buffers are written for the last ResBlock's output and convPa.3's.  sfd2_debug_activation must say so instead of handing
            // back what an earlier sfd2_det left in them
            for (auto &kv : c->acts)
                if (kv.first != "conv4.2" && kv.first != "convPa" && kv.first != "convPb" && kv.first != "convDb" && kv.first != "ConvSta")
                    kv.second.absent = true;
        }
        return 0;
    }
    if (!layers) return 0;   // throughput path: intermediates live in aliased arena slots and are not readable
    if (!fused_stem) reg("conv1a", c->a1a.p, 0, 0, 64, 64, H, W);
    reg("bn1b", c->a1b.p, 0, 0, 64, 64, c->H2, c->W2);
    reg("conv2a", c->a2a.p, 0, 0, 128, 128, c->H2, c->W2);
    reg("bn2b", c->a2b.p, 0, 0, 128, 128, c->H4, c->W4);
    reg("conv3a", c->a3a.p, 0, 0, 256, 256, c->H4, c->W4);
    reg("bn3b", c->a3b.p, 0, 0, 256, 256, c->H4, c->W4);
    for (int b = 0; b < 3; ++b) {
        if (!fused_rb) {
            reg(n1[b], c->rt1[b].p, 0, 0, 256, 256, c->H4, c->W4);
            reg(n2[b], c->rt2[b].p, 0, 0, 256, 256, c->H4, c->W4);
        }
        reg(n3[b], c->ro[b].p, 0, 0, 256, 256, c->H4, c->W4);
    }
    reg("convPa.0", c->pa0_o.p, 0, 0, 256, 256, c->H8, c->W8);
    reg("convPa", c->pa_o.p, 0, 0, 256, 256, c->H8, c->W8);
    reg("convDa.0", c->da0_o.p, 0, 0, 256, 256, c->H4, c->W4);
    reg("convDa", c->da_o.p, 0, 0, 256, 256, c->H4, c->W4);
    return 0;
}

static void conv(sfd2_ctx *c, const char *name, const ConvW &L, const DevPtr &in, int H, int W, const DevPtr &out,
                 int Ho, int Wo, int relu, const half_t *res = nullptr, int out_f32 = 0, const float *scale_override = nullptr)
{
    char kn[48];
    const float *scale = scale_override ? scale_override : L.scale.as<float>();
    const int bn = (L.cout_pad % 256 == 0) ? 256 : (L.cout_pad % 128 == 0 ? 128 : 64);
    snprintf(kn, sizeof(kn), "conv_igemm<%d,%d,%d%s>", L.ks, L.stride, bn, out_f32 ? ",f32" : "");
    if (!res && !out_f32 && conv3x3_pp_serves(L.ks, L.stride, L.cout_pad, L.cin)) snprintf(kn, sizeof(kn), "conv3x3_pp");
    if (!res && !out_f32 && conv3x3_rf_serves(L.ks, L.stride, L.cout_pad, L.cin, Ho, Wo)) snprintf(kn, sizeof(kn), "conv3x3_rf<%d>", L.stride);
    const double px = (double)Ho * Wo;
    const double flops = 2.0 * px * L.cout * L.cin * L.ks * L.ks;
    const double bytes = 2.0 * ((double)H * W * L.cin + (double)L.cout * L.cin * L.ks * L.ks) +
                         px * L.cout_pad * (out_f32 ? 4.0 : 2.0) + (res ? px * L.cout_pad * 2.0 : 0.0);
    static const bool no_c1 = sfd2_env("SFD2_NO_CONV1X1") != nullptr;
    if (L.wrm.p && !out_f32 && !no_c1) {
        snprintf(kn, sizeof(kn), "conv1x1_c256%s", res ? "+res" : "");
        ProfScope ps(c, name, kn, flops, bytes);
        launch_conv1x1_c256(c->cur_stream, in.as<half_t>(), Ho * Wo, L.wrm.as<half_t>(), scale, L.shift.as<float>(),
                            relu, res, reinterpret_cast<half_t *>(out.p), c->zero_page.as<half_t>());
        return;
    }
    ProfScope ps(c, name, kn, flops, bytes);
    launch_conv_igemm(c->cur_stream, in.as<half_t>(), H, W, L.cin, L.w.as<half_t>(), scale,
                      L.shift.as<float>(), L.cout_pad, L.ks, L.stride, relu, res, out.p, out_f32, Ho, Wo,
                      c->zero_page.as<half_t>());
}

// SFD2_PREC_F16C: one compensated layer.  A compensated tensor = hi plane followed by its corr plane; in_comp / out_comp
// say which of the two tensors have one (a plain-fp16 consumer just reads the hi plane).
// the range-status slot of a stored tensor of the compensated mode (sfd2_internal.h), null when nothing is recorded
static unsigned int *range_slot(sfd2_ctx *c, int id)
{
    return (id >= 0 && c->range_stat.p) ? c->range_stat.as<unsigned int>() + id * SFD2_RANGE_SUB : nullptr;
}
static half_t *corr_of(const DevPtr &b, size_t px, int pitch) { return b.as<half_t>() + px * (size_t)pitch; }
static void convc(sfd2_ctx *c, const char *name, const ConvW &L, const DevPtr &in, int H, int W, const DevPtr &out,
                  int Ho, int Wo, int relu, bool in_comp, bool out_comp, const DevPtr *res = nullptr, int rs_id = -1 /* SFD2_RS_*: the output's range-status slot */,
                  int fmt6 = 0 /* option "fp6_acts": bit 0 = the input's corr records are fp6 half-records, bit 1 = the output's are to be */)
{
    unsigned int *rs = range_slot(c, rs_id);
    char kn[48];
    snprintf(kn, sizeof(kn), "convc_igemm<%d,%d>", L.ks, L.stride);
    const double px = (double)Ho * Wo;
    const double flops = 2.0 * px * L.cout * L.cin * L.ks * L.ks;
    const double bytes = (in_comp ? 4.0 : 2.0) * ((double)H * W * L.cin + (double)L.cout * L.cin * L.ks * L.ks) +
                         px * L.cout_pad * (out_comp ? 4.0 : 2.0) + (res ? px * L.cout_pad * 4.0 : 0.0);
    const half_t *in_c = in_comp ? corr_of(in, (size_t)H * W, L.cin) : nullptr;
    half_t *out_c = out_comp ? corr_of(out, (size_t)Ho * Wo, L.cout_pad) : nullptr;
    // conv3x3_pp's tile is 128 channels wide: in its compensated form it also takes conv2a (64 -> 128 channels, four chunks)
    if (!res && !c->opt_generic_c && L.ks == 3 && L.stride == 1 && L.cout_pad % 128 == 0 && L.cin % 64 == 0) {
        ProfScope ps(c, name, (in_c && out_c) ? "conv3x3_pp<comp>" : (in_c ? "conv3x3_pp<comp,plain out>" : "conv3x3_pp<comp out>"), flops, bytes);
        if (relu && ((!in_c && out_c) || (in_c && !out_c && (fmt6 & 1) && L.wc66.p && L.sa66.p))) {
            // option "c3b_plain": conv3b over the hi plane of its input alone (the hi chunks at the start of any of the layer's arrays; the output's corr
            // bytes from the fp32 accumulators as ever: fmt6 bit 3 = the three-byte trunk form), and conv3a, whose corr plane then has no reader
            launch_conv3x3_pp_c(c->cur_stream, in.as<half_t>(), in_c, H, W, L.cin, in_c ? L.wc66.as<half_t>() : L.wc.as<half_t>(), L.scale.as<float>(),
                                in_c ? L.shift.as<float>() : L.shift.as<float>(), L.cout_pad, relu, out.as<half_t>(), out_c, Ho, Wo, c->zero_page.as<half_t>(), L.sbyte,
                                in_c ? L.sa66.as<float>() : nullptr, rs, fmt6);
            return;
        }
        if (fmt6 && in_c && out_c && relu && L.wc66.p && L.sa66.p && L.wc6.p && L.sa6.p) {
            // fp6 pixel records on either side: the filter strings in the input records' format (fp6 x fp6 when the input's are fp6)
            const bool i6 = (fmt6 & 1) != 0;
            launch_conv3x3_pp_c(c->cur_stream, in.as<half_t>(), in_c, H, W, L.cin, i6 ? L.wc66.as<half_t>() : L.wc6.as<half_t>(), L.scale.as<float>(),
                                L.shift.as<float>(), L.cout_pad, relu, out.as<half_t>(), out_c, Ho, Wo, c->zero_page.as<half_t>(), L.sbyte,
                                i6 ? L.sa66.as<float>() : L.sa6.as<float>(), rs, fmt6);      // (bit 2: the output stored space-to-depth, with fp6 input records only)
            return;
        }
        if (fmt6) { c->net_error = 1; (void)fail(std::string(name) + ": fp6 records requested from a layer without fp6 filter strings"); return; }
        const bool f6 = c->opt_fp6_filters && in_c && out_c && L.wc6.p && L.sa6.p;      // corr filters as fp6 (option "fp6_filters")
        launch_conv3x3_pp_c(c->cur_stream, in.as<half_t>(), in_c, H, W, L.cin, f6 ? L.wc6.as<half_t>() : L.wc.as<half_t>(), L.scale.as<float>(),
                            L.shift.as<float>(), L.cout_pad, relu, out.as<half_t>(), out_c, Ho, Wo, c->zero_page.as<half_t>(), L.sbyte,
                            f6 ? L.sa6.as<float>() : nullptr, rs);
        return;
    }
    if ((fmt6 & 1) || ((fmt6 & 2) && !(!c->opt_generic_c && !c->opt_no_rf_c && conv3x3_rf_c_serves(L.ks, L.stride, L.cout_pad, L.cin, Ho, Wo) && relu))) {
        // (run_network decides the format per tensor from the same predicates: reaching this is a bug, reported as an error of the call)
        c->net_error = 1;
        (void)fail(std::string(name) + ": fp6 records on a path that cannot read / write them");
        return;
    }
    if (!c->opt_generic_c && !c->opt_no_rf_c && in_c && out_c && !res && L.ks == 3 && L.stride == 2 && L.cout_pad == 128) {   // conv2b
        ProfScope ps(c, name, "conv3x3_rf<2,comp>", flops, bytes);
        if (!launch_conv3x3_rf_c(c->cur_stream, in.as<half_t>(), in_c, H, W, L.cin, L.wc.as<half_t>(), L.scale.as<float>(),
                                L.shift.as<float>(), L.cout_pad, L.stride, relu, out.as<half_t>(), out_c, Ho, Wo,
                                c->zero_page.as<half_t>(), L.sbyte, rs, fmt6 & 2))
            ps.cancel();     // no instantiation for this geometry: the next candidate takes the layer (and the profile row)
        else
            return;
    }
    if (!c->opt_generic_c && in_c && out_c && L.wfh.p && L.wfc.p) {   // the ResBlocks' 1x1 layers: persistent streaming kernel
        snprintf(kn, sizeof(kn), "conv1x1_c256<comp>%s", res ? "+res" : "");
        ProfScope ps(c, name, kn, flops, bytes);
        launch_conv1x1_c256_c(c->cur_stream, in.as<half_t>(), in_c, Ho * Wo, L.wfh.as<half_t>(), L.wfc.as<half_t>(), L.scale.as<float>(),
                              L.shift.as<float>(), relu, res ? res->as<half_t>() : nullptr,
                              res ? corr_of(*res, (size_t)Ho * Wo, L.cout_pad) : nullptr, out.as<half_t>(), out_c,
                              c->zero_page.as<half_t>(), L.sbyte, rs);
        return;
    }
    if (!c->opt_generic_c && in_c && out_c && L.cout_pad % 128 == 0 && ((L.ks == 1 && L.stride == 1) || (L.ks == 3 && L.stride == 2 && !res))) {
        snprintf(kn, sizeof(kn), "conv_igemm2<%d,%d,comp>%s", L.ks, L.stride, res ? "+res" : "");
        ProfScope ps(c, name, kn, flops, bytes);
        if (!launch_conv_igemm2_c(c->cur_stream, in.as<half_t>(), in_c, H, W, L.cin, L.wc.as<half_t>(), L.scale.as<float>(),
                                 L.shift.as<float>(), L.cout_pad, L.ks, L.stride, relu, res ? res->as<half_t>() : nullptr,
                                 res ? corr_of(*res, (size_t)Ho * Wo, L.cout_pad) : nullptr, out.as<half_t>(), out_c, Ho, Wo,
                                 c->zero_page.as<half_t>(), L.sbyte, rs))
            ps.cancel();
        else
            return;
    }
    snprintf(kn, sizeof(kn), "convc_igemm<%d,%d>", L.ks, L.stride);
    ProfScope ps(c, name, kn, flops, bytes);
    launch_convc_igemm(c->cur_stream, in.as<half_t>(), in_comp ? corr_of(in, (size_t)H * W, L.cin) : nullptr, H, W, L.cin,
                       L.wc.as<half_t>(), L.scale.as<float>(), L.shift.as<float>(), L.cout_pad, L.ks, L.stride, relu,
                       res ? res->as<half_t>() : nullptr, res ? corr_of(*res, (size_t)Ho * Wo, L.cout_pad) : nullptr,
                       out.as<half_t>(), out_comp ? corr_of(out, (size_t)Ho * Wo, L.cout_pad) : nullptr, Ho, Wo, L.sbyte, rs);
}

// which throughput kernel takes a layer of SFD2_PREC_F16X3 from hi / lo' planes: 0 none (generic, fp32 in / out), 1 conv3x3_pp, 2 conv3x3_rf
// (small outputs and stride 2: conv2b, convPa.0 / convPa.3 at 1600x1200, every 256-channel layer of a 640x480 image -- as in f16)
static int x3_fast_kind(const sfd2_ctx *c, const ConvW &L, bool has_res, int Ho, int Wo)
{
    if (c->precision != SFD2_PREC_F16X3 || !c->opt_x3_pp || has_res || L.ks != 3 || L.cin % 64 != 0) return 0;
    if ((L.cout_pad == 256 || (L.cout_pad == 128 && L.stride == 2)) && conv3x3_rf_serves(3, L.stride, L.cout_pad, L.cin, Ho, Wo)) return 2;
    return (L.stride == 1 && L.cout_pad % 128 == 0) ? 1 : 0;
}

int convf(sfd2_ctx *c, const char *name, const ConvW &L, const DevPtr &in, int H, int W, const DevPtr &out,
          int Ho, int Wo, int relu, const float *res)
{
    char kn[64];
    const bool x3 = c->precision == SFD2_PREC_F16X3;
    const int kind = x3_fast_kind(c, L, res != nullptr, Ho, Wo);
    const bool use_rf = kind == 2;
    if (kind != 0) {
        // The 3x3 layers (half of this mode's time) on the throughput kernels: the input is split ONCE into hi / lo'
        // planes (the generic kernel splits every staged piece, per tile and chunk), conv3x3_pp / conv3x3_rf stage the planes by direct
        // copies and run their fp16 K loop three times (hi x hi, hi x lo', lo' x hi) into one accumulator; fp32 or planes out.
        ConvW &Lm = const_cast<ConvW &>(L);
        const size_t nfl = (size_t)L.ks * L.ks * L.cout_pad * L.cin, nin = (size_t)H * W * L.cin;
        if (!L.wx3p.p) {
            if (Lm.wx3p.ensure(nfl * 2 * sizeof(half_t)) != hipSuccess) return fail("out of device memory (f16x3 filter planes)");
            launch_x3_split_planes(c->stream, L.w.as<float>(), nfl, Lm.wx3p.p, Lm.wx3p.as<half_t>() + nfl);
        }
        const half_t *ph = nullptr, *pl = nullptr;
        const bool pre = c->x3_pre_src == in.p && c->x3_pre_hi;     // the producer left the planes behind: no split
        if (pre) { ph = c->x3_pre_hi; pl = c->x3_pre_lo; c->x3_pre_src = nullptr; }
        else {
            if (c->x3_planes.ensure(nin * 2 * sizeof(half_t)) != hipSuccess) return fail("out of device memory (f16x3 activation planes)");
            ph = c->x3_planes.as<half_t>(); pl = ph + nin;
        }
        const size_t nout = (size_t)Ho * Wo * L.cout_pad;
        half_t *oh = nullptr, *ol = nullptr;
        if (c->x3_planes_out_now) {      // planes out (the only reader is the next 3x3 layer / the ResBlocks / the sparse descriptor head)
            DevBuf &dst = c->x3_planes_out_now == 2 ? c->x3_da0_planes : c->x3_planes_out_now == 3 ? c->x3_rb_planes[0]
                          : (ph == c->x3_chain.as<half_t>() ? c->x3_chain2 : c->x3_chain);
            if (dst.ensure(nout * 2 * sizeof(half_t)) != hipSuccess) return fail("out of device memory (f16x3 activation planes)");
            oh = dst.as<half_t>(); ol = oh + nout;
            c->x3_pre_src = out.p; c->x3_pre_hi = oh; c->x3_pre_lo = ol;
        }
        snprintf(kn, sizeof(kn), "%sconv3x3_%s<x3%s>", pre ? "" : "x3_split_planes + ", use_rf ? "rf" : "pp", oh ? ", planes out" : "");
        ProfScope ps(c, name, kn, 2.0 * (double)Ho * Wo * L.cout * L.cin * 9, (pre ? 4.0 : 12.0) * nin + 4.0 * nout);
        if (!pre) launch_x3_split_planes(c->stream, in.as<float>(), nin, const_cast<half_t *>(ph), const_cast<half_t *>(pl));
        if (use_rf && launch_conv3x3_rf_x3(c->stream, ph, pl, H, W, L.cin, L.wx3p.as<half_t>(), L.scale.as<float>(), L.shift.as<float>(), L.cout_pad,
                                           L.stride, relu, oh, ol, oh ? nullptr : out.as<float>(), Ho, Wo, c->zero_page.as<half_t>()))
            return 0;
        if (L.stride != 1) { ps.cancel(); return fail("conv3x3_rf<x3>: no instantiation for this layer (x3_fast_kind and the launcher disagree)"); }
        launch_conv3x3_pp_x3(c->stream, ph, pl, H, W, L.cin, L.wx3p.as<half_t>(), L.scale.as<float>(), L.shift.as<float>(), L.cout_pad, relu,
                             oh, ol, oh ? nullptr : out.as<float>(), Ho, Wo, c->zero_page.as<half_t>(), (oh && c->x3_s2d_out_now) ? 1 : 0);
        return 0;
    }
    if (x3 && !L.wx3.p) {      // split the packed filters once: the kernel then stages them without arithmetic
        ConvW &Lm = const_cast<ConvW &>(L);
        const size_t nfl = (size_t)L.ks * L.ks * L.cout_pad * L.cin;
        if (Lm.wx3.ensure(nfl * sizeof(float)) != hipSuccess) return fail("out of device memory (f16x3 filters)");
        launch_x3_split(c->stream, L.w.as<float>(), nfl, Lm.wx3.p);
    }
    snprintf(kn, sizeof(kn), "conv_igemm_%s<%d,%d,%d>", x3 ? "x3" : "f32", L.ks, L.stride, (L.cout_pad % 128 == 0) ? 128 : 64);
    const double px = (double)Ho * Wo;
    ProfScope ps(c, name, kn, 2.0 * px * L.cout * L.cin * L.ks * L.ks,
                 4.0 * ((double)H * W * L.cin + (double)L.cout * L.cin * L.ks * L.ks + px * L.cout_pad * (res ? 2 : 1)));
    (x3 ? launch_conv_igemm_x3 : launch_conv_igemm_f32)(c->stream, in.as<float>(), H, W, L.cin, x3 ? L.wx3.as<float>() : L.w.as<float>(), L.scale.as<float>(),
                          L.shift.as<float>(), L.cout_pad, L.ks, L.stride, relu, res, out.as<float>(), Ho, Wo);
    return 0;
}

// strict mode: identical layer sequence on fp32 activations (conv_f32_kernels.hip)
static int run_network_f32(sfd2_ctx *c, const float *img_dev, int normalise)
{
    hipStream_t st = c->stream;
    const int H = c->H, W = c->W, H2 = c->H2, W2 = c->W2, H4 = c->H4, W4 = c->W4, H8 = c->H8, W8 = c->W8;
    const double P1 = (double)H * W, P4 = (double)H4 * W4, P8 = (double)H8 * W8;
    c->x3_pre_src = nullptr;           // (no planes of an earlier pass are left over)
    c->x3_planes_out_now = 0;
    if (c->x3_fast_rb_now && c->w1b_stem_x3.p && c->c1a.wc.p) {
        // throughput path of SFD2_PREC_F16X3: the fused stem in three-pass arithmetic (conv1a's image and filters as hi + lo fp16 as
        // in the compensated mode, conv1b on hi / lo' planes of conv1a's tile in LDS), its output as planes for conv2a
        const size_t nout = (size_t)H2 * W2 * 64;
        HIPCHECK(c->x3_chain.ensure(std::max(nout, (size_t)H4 * W4 * 256) * 2 * sizeof(half_t)));
        ProfScope ps(c, "conv1a+conv1b", "fused_stem_c_kernel<x3>", 2.0 * P1 * 64 * 27 + 2.0 * (double)H2 * W2 * 64 * 576, P1 * 12 + (double)H2 * W2 * 256);
        launch_fused_stem_c(st, img_dev, H, W, normalise, c->c1a.wc.as<half_t>(), c->f1a.scale.as<float>(), c->f1a.shift.as<float>(),
                            c->w1b_stem_x3.p, c->f1b.scale.as<float>(), c->f1b.shift.as<float>(), c->x3_chain.as<half_t>(),
                            c->x3_chain.as<half_t>() + nout, H2, W2, -1);
        c->x3_pre_src = c->g1b.p; c->x3_pre_hi = c->x3_chain.as<half_t>(); c->x3_pre_lo = c->x3_chain.as<half_t>() + nout;
    } else {
    {
        ProfScope ps(c, "conv1a", "conv1a_f32_kernel", 2.0 * P1 * 64 * 27, P1 * (12 + 256));
        launch_conv1a_f32(st, img_dev, H, W, normalise, c->f1a.w.as<float>(), c->f1a.scale.as<float>(),
                          c->f1a.shift.as<float>(), c->g1a.as<float>());
    }
    if (convf(c, "conv1b", c->f1b, c->g1a, H, W, c->g1b, H2, W2, 1)) return -1;
    }
    // (throughput path: a layer whose only reader takes planes writes planes and no fp32 tensor: conv2a -> conv2b -> conv3a -> conv3b ->
    // ResBlocks, convDa.0 -> the sparse descriptor head)
    const bool fast_rb = c->x3_fast_rb_now && c->rb1[0].wfh.p && c->rb1[0].wfl.p && c->rb2[0].w.p && c->rb2[0].wlk.p;
    const bool k2b = c->x3_fast_rb_now && x3_fast_kind(c, c->f2b, false, H4, W4) != 0, k3a = c->x3_fast_rb_now && x3_fast_kind(c, c->f3a, false, H4, W4) != 0;
    const bool k3b = c->x3_fast_rb_now && x3_fast_kind(c, c->f3b, false, H4, W4) != 0;
    // Option "s2d" in this mode too (round 5): conv2a (conv3x3_pp<x3>) stores its planes space-to-depth and conv2b runs as a stride-1 layer over
    // them (conv2b_s2d_kernel<x3>) instead of on the strided conv3x3_rf<2, x3>; planes out for conv3a
    const bool s2d_x3 = c->opt_s2d && k2b && k3a && x3_fast_kind(c, c->f2a, false, H2, W2) == 1 && conv2b_s2d_serves(H2, W2, c->f2b.cin, c->f2b.cout_pad) &&
                        H4 * 2 == H2 && W4 * 2 == W2;
    c->x3_planes_out_now = k2b ? 1 : 0;
    c->x3_s2d_out_now = s2d_x3 ? 1 : 0;
    if (convf(c, "conv2a", c->f2a, c->g1b, H2, W2, c->g2a, H2, W2, 1)) return -1;
    c->x3_s2d_out_now = 0;
    c->x3_planes_out_now = k3a ? 1 : 0;
    if (s2d_x3 && c->x3_pre_src == c->g2a.p && c->x3_pre_hi) {
        ConvW &L = c->f2b;
        const size_t nfl = (size_t)9 * L.cout_pad * L.cin, nout = (size_t)H4 * W4 * L.cout_pad;
        if (!L.wx3p.p) {
            HIPCHECK(L.wx3p.ensure(nfl * 2 * sizeof(half_t)));
            launch_x3_split_planes(st, L.w.as<float>(), nfl, L.wx3p.p, L.wx3p.as<half_t>() + nfl);
        }
        const half_t *ph = c->x3_pre_hi, *pl = c->x3_pre_lo;
        DevBuf &dst = (ph == c->x3_chain.as<half_t>()) ? c->x3_chain2 : c->x3_chain;
        HIPCHECK(dst.ensure(nout * 2 * sizeof(half_t)));
        half_t *oh = dst.as<half_t>(), *ol = oh + nout;
        {
            ProfScope ps(c, "conv2b", "conv2b_s2d_kernel<x3>", 2.0 * (double)H4 * W4 * L.cout * L.cin * 9, 4.0 * (double)H2 * W2 * L.cin + 4.0 * nout);
            launch_conv2b_s2d_x3(st, ph, pl, H4, W4, L.wx3p.as<half_t>(), L.scale.as<float>(), L.shift.as<float>(), 1, oh, ol, c->zero_page.as<half_t>());
        }
        c->x3_pre_src = c->g2b.p; c->x3_pre_hi = oh; c->x3_pre_lo = ol;
    } else if (convf(c, "conv2b", c->f2b, c->g2a, H2, W2, c->g2b, H4, W4, 1)) return -1;
    c->x3_planes_out_now = k3b ? 1 : 0;
    if (convf(c, "conv3a", c->f3a, c->g2b, H4, W4, c->g3a, H4, W4, 1)) return -1;
    c->x3_planes_out_now = (fast_rb && k3b) ? 3 : 0;
    if (convf(c, "conv3b", c->f3b, c->g3a, H4, W4, c->g3b, H4, W4, 1)) return -1;
    c->x3_planes_out_now = 0;
    const bool rb_in_planes = c->x3_pre_src == c->g3b.p;      // conv3b left the ResBlocks' input planes in x3_rb_planes[0]
    c->x3_pre_src = nullptr;
    const DevPtr *x = &c->g3b;
    static const char *nm1[3] = {"conv4.0.conv1", "conv4.1.conv1", "conv4.2.conv1"};
    static const char *nm2[3] = {"conv4.0.conv2", "conv4.1.conv2", "conv4.2.conv2"};
    static const char *nm3[3] = {"conv4.0.conv3", "conv4.1.conv3", "conv4.2.conv3"};
    if (fast_rb) {
        // ResBlocks of SFD2_PREC_F16X3 on the throughput path: every tensor of a block lives as hi / lo' planes (the input is split
        // once in front of the first block).  conv1 and conv3 on the streaming three-pass 1x1 kernel (filters = the fp16 set's
        // fragment-ordered hi / lo' arrays: the same split of the same fp32 weights), the grouped conv on gconv_c_kernel<X3>
        // (pre-split operands, direct plane stores); the skip connection is read from the planes (22 significant bits) and only the
        // last block also writes the fp32 tensor its generic readers (convPa.0, ConvSta) take.
        const size_t nin = (size_t)H4 * W4 * 256;
        for (int k = 0; k < 3; ++k) HIPCHECK(c->x3_rb_planes[k].ensure(nin * 2 * sizeof(half_t)));
        half_t *xh = c->x3_rb_planes[0].as<half_t>(), *xl = xh + nin, *th = c->x3_rb_planes[1].as<half_t>(), *tl = th + nin;
        half_t *uh = c->x3_rb_planes[2].as<half_t>(), *ul = uh + nin;
        if (!rb_in_planes) {
            ProfScope ps(c, "conv3b planes", "x3_split_planes", 0.0, 12.0 * nin);
            launch_x3_split_planes(st, x->as<float>(), nin, xh, xl);
        }
        for (int b = 0; b < 3; ++b) {
            {
                ProfScope ps(c, nm1[b], "conv1x1_c256<x3>", 2.0 * P4 * 256.0 * 256.0, P4 * 256.0 * 8);
                launch_conv1x1_c256_x3(st, xh, xl, H4 * W4, c->rb1[b].wfh.as<half_t>(), c->rb1[b].wfl.as<half_t>(), c->frb1[b].scale.as<float>(),
                                       c->frb1[b].shift.as<float>(), 1, nullptr, nullptr, nullptr, th, tl, c->zero_page.as<half_t>());
            }
            {
                ProfScope ps(c, nm2[b], "gconv_c_kernel<x3>", 2.0 * P4 * 256 * 72, P4 * 256 * 8);
                launch_gconv_c(st, th, tl, H4, W4, c->rb2[b].w.as<half_t>(), c->rb2[b].wlk.p, c->frb2[b].scale.as<float>(),
                               c->frb2[b].shift.as<float>(), uh, ul, -1, 0, H4);
            }
            {
                ProfScope ps(c, nm3[b], "conv1x1_c256<x3>+res", 2.0 * P4 * 256.0 * 256.0, P4 * 256.0 * (b == 2 ? 16 : 12));
                launch_conv1x1_c256_x3(st, uh, ul, H4 * W4, c->rb3[b].wfh.as<half_t>(), c->rb3[b].wfl.as<half_t>(), c->frb3[b].scale.as<float>(),
                                       c->frb3[b].shift.as<float>(), 1, xh, xl, b == 2 ? c->gro[b].as<float>() : nullptr, xh, xl,
                                       c->zero_page.as<half_t>());
            }
        }
        x = &c->gro[2];
        c->x3_pre_src = x->p; c->x3_pre_hi = xh; c->x3_pre_lo = xl;     // the backbone output's planes: convDa.0 takes them as they are
    }
    for (int b = 0; b < (fast_rb ? 0 : 3); ++b) {
        if (convf(c, nm1[b], c->frb1[b], *x, H4, W4, c->grt1[b], H4, W4, 1)) return -1;
        {
            if (c->precision == SFD2_PREC_F16X3) {
                if (!c->frb2[b].wx3.p) {
                    HIPCHECK(c->frb2[b].wx3.ensure((size_t)16 * 5 * 64 * 16 * sizeof(half_t)));
                    launch_gconv_x3_pack(st, c->frb2[b].w.as<float>(), c->frb2[b].wx3.p);
                }
                ProfScope ps(c, nm2[b], "gconv_x3_kernel", 2.0 * P4 * 256 * 72, P4 * 256 * 8);
                launch_gconv_x3(st, c->grt1[b].as<float>(), H4, W4, c->frb2[b].wx3.p, c->frb2[b].scale.as<float>(),
                                c->frb2[b].shift.as<float>(), c->grt2[b].as<float>());
            } else {
                ProfScope ps(c, nm2[b], "gconv_f32_kernel", 2.0 * P4 * 256 * 72, P4 * 256 * 8);
                launch_gconv_f32(st, c->grt1[b].as<float>(), H4, W4, c->frb2[b].w.as<float>(), c->frb2[b].scale.as<float>(),
                                 c->frb2[b].shift.as<float>(), c->grt2[b].as<float>());
            }
        }
        if (convf(c, nm3[b], c->frb3[b], c->grt2[b], H4, W4, c->gro[b], H4, W4, 1, x->as<float>())) return -1;
        x = &c->gro[b];
    }
    // (throughput path: convPa.0 reads the backbone output's planes and leaves its own output as planes for convPa.3)
    const void *bb_src = c->x3_pre_src;
    const half_t *bb_hi = c->x3_pre_hi, *bb_lo = c->x3_pre_lo;
    c->x3_planes_out_now = c->x3_fast_rb_now ? 1 : 0;
    if (convf(c, "convPa.0", c->fpa0, *x, H4, W4, c->gpa0_o, H8, W8, 1)) return -1;
    c->x3_planes_out_now = 0;
    if (convf(c, "convPa.3", c->fpa3, c->gpa0_o, H8, W8, c->gpa_o, H8, W8, 0)) return -1;
    if (convf(c, "convPb", c->fpb, c->gpa_o, H8, W8, c->logits, H8, W8, 0)) return -1;
    c->x3_pre_src = bb_src; c->x3_pre_hi = bb_hi; c->x3_pre_lo = bb_lo;      // convDa.0 takes the same planes
    const bool da0_planes = c->skip_da3_now && c->x3_fast_rb_now;
    // Option "x3_desc16": the descriptor branch leaves this mode's arithmetic here -- convDa.0 as ONE fp16 pass (conv3x3_pp) over the backbone output's hi
    // plane, at the network's own scale (scale_rawin), its output (times 2^act_exp[AE_DA0], what convDa.3's folded constants expect) in the buffer
    // the planes would have taken; sfd2_extract then runs the fp16 sparse head on it.  The detector branch above is untouched.
    c->x3_desc16_now = 0;
    if (da0_planes && c->opt_x3_desc16 && bb_src == x->p && bb_hi && c->da0.w.p && c->da0.scale_rawin.p && c->da3.w.p && c->db.w.p) {
        const size_t nin = (size_t)H4 * W4 * 256;
        HIPCHECK(c->x3_da0_planes.ensure(nin * 2 * sizeof(half_t)));
        DevPtr in, out;
        in.p = const_cast<half_t *>(bb_hi);
        out.p = c->x3_da0_planes.p;
        conv(c, "convDa.0", c->da0, in, H4, W4, out, H4, W4, 1, nullptr, 0, c->da0.scale_rawin.as<float>());
        c->da0_cur = c->x3_da0_planes.as<half_t>();
        c->x3_desc16_now = 1;
    } else {
    c->x3_planes_out_now = da0_planes ? 2 : 0;
    if (convf(c, "convDa.0", c->fda0, *x, H4, W4, c->gda0_o, H4, W4, 1)) return -1;
    }
    c->x3_planes_out_now = 0;
    c->x3_pre_src = nullptr;
    if (c->x3_desc16_now) {
        // (nothing: convDa.3 and convDb run on the sampled corners, in fp16)
    } else if (c->skip_da3_now) {      // sparse descriptor head of SFD2_PREC_F16X3 (sfd2_extract): convDa.3 and convDb run on the sampled corners only
        const size_t nin = (size_t)H4 * W4 * 256;
        if (!da0_planes) {
            HIPCHECK(c->x3_da0_planes.ensure(nin * 2 * sizeof(half_t)));
            ProfScope ps(c, "convDa.0 planes", "x3_split_planes", 0.0, 12.0 * nin);
            launch_x3_split_planes(st, c->gda0_o.as<float>(), nin, c->x3_da0_planes.p, c->x3_da0_planes.as<half_t>() + nin);
        }
    } else {
        if (convf(c, "convDa.3", c->fda3, c->gda0_o, H4, W4, c->gda_o, H4, W4, 0)) return -1;
        if (convf(c, "convDb", c->fdb, c->gda_o, H4, W4, c->draw, H4, W4, 0)) return -1;
    }
    if (c->has_sta) {
        ProfScope ps(c, "ConvSta", "convsta_f32_kernel", 2.0 * P4 * 3 * 256, P4 * (1024 + 12));
        launch_convsta_f32(st, x->as<float>(), H4 * W4, c->sta_w.as<float>(), c->sta_b.as<float>(), c->sta.as<float>());
    }
    if (!c->skip_head_now) {
        ProfScope ps(c, "detector_head", "detector_head_kernel", 0.0, P8 * (65 * 4 + 256));
        launch_detector_head(st, c->logits.as<float>(), 128, H8, W8, c->score.as<float>());
    }
    HIPCHECK(hipGetLastError());
    return 0;
}

// ResSegNetV2.det up to the three head outputs (nets/sfd2.py:314-328, :340-345)
int run_network(sfd2_ctx *c, const float *img_dev, int normalise)
{
    c->cur_stream = c->stream;
    g_sfd2_cu_limit = c->opt_cu_limit;
    if (c->precision == SFD2_PREC_F32 || c->precision == SFD2_PREC_F16X3) return run_network_f32(c, img_dev, normalise);
    const bool comp = c->precision == SFD2_PREC_F16C;
    hipStream_t st = c->stream;
    const int H = c->H, W = c->W, H2 = c->H2, W2 = c->W2, H4 = c->H4, W4 = c->W4, H8 = c->H8, W8 = c->W8;
    const double P1 = (double)H * W, P4 = (double)H4 * W4, P8 = (double)H8 * W8;
    // Activation placement.  det (the parity entry point) keeps every activation in its own buffer for
    // sfd2_debug_activation.  The throughput path (sfd2_extract) packs the whole chain into three 61 MB slots of one
    // arena (at 1600x1200), reusing a slot as soon as its tensor is dead, so the working set fits the 256 MB
    // Infinity Cache and a layer mostly reads what the previous one just wrote (measured: ResBlocks 400 -> 344 us;
    // four slots measure the same as three).
    const bool alias = c->alias_now != 0;
    DevPtr a1b = c->a1b, a2a = c->a2a, a2b = c->a2b, a3a = c->a3a, a3b = c->a3b, pa0_o = c->pa0_o, pa_o = c->pa_o,
           da0_o = c->da0_o, da_o = c->da_o;   // non-owning views
    DevPtr t1v[3] = {c->rt1[0], c->rt1[1], c->rt1[2]}, t2v[3] = {c->rt2[0], c->rt2[1], c->rt2[2]},
           rov[3] = {c->ro[0], c->ro[1], c->ro[2]};
    if (alias) {
        const size_t P2 = (size_t)H2 * W2, P4s = (size_t)H4 * W4, P8s = (size_t)H8 * W8;
        size_t S = std::max(P2 * 64 * 2, P4s * 256 * 2);
        S = std::max(S, (P2 * 128 * 2 + 1) / 2);
        S = std::max(S, 2 * (P8s * 256 * 2 + 256));
        if (comp) S *= 2;   // hi plane + corr plane per tensor (the corr plane follows the hi plane inside the slot)
        S = (S + 255) & ~(size_t)255;
        HIPCHECK(c->arena.ensure((c->opt_branches ? 4 : 3) * S));
        char *base = c->arena.as<char>();
        auto slot = [&](int i, size_t off = 0) { DevPtr v; v.p = base + (size_t)i * S + off; v.cap = 0; return v; };
        {
            // three slots (184 MB): a ResBlock's output overwrites its own t1 (dead once conv3 starts), the next
            // block's t1 takes the slot of the previous input
            a1b = slot(0); a2a = slot(1) /* spans slots 1-2 */; a2b = slot(0); a3a = slot(1); a3b = slot(2);
            t1v[0] = slot(0); t2v[0] = slot(1); rov[0] = slot(0);    // x = slot 2
            t1v[1] = slot(2); t2v[1] = slot(1); rov[1] = slot(2);    // x = slot 0
            t1v[2] = slot(0); t2v[2] = slot(1); rov[2] = slot(0);    // x = slot 2 -> final x = slot 0
            if (comp && c->opt_comp_rb && !c->opt_generic_c && c->opt_rb_inner >= 2 && c->opt_fuse_rb23) {
                // rb23_c_kernel reads t1 (with the halo rows of neighbouring tiles) while other tiles already write the block's
                // output: the output cannot take t1's slot.  No t2 in HBM on this path, so three slots still do.
                t1v[0] = slot(0); rov[0] = slot(1);    // x = slot 2
                t1v[1] = slot(0); rov[1] = slot(2);    // x = slot 1
                t1v[2] = slot(1); rov[2] = slot(0);    // x = slot 2 -> final x = slot 0
            }
            pa0_o = slot(1); pa_o = slot(1, (P8s * 256 * 2 * (comp ? 2 : 1) + 255) & ~(size_t)255);   // (convPa.0's corr plane with "comp_heads")
            da0_o = slot(2); da_o = slot(1);   // convDa.3 runs after convPb has consumed slot 1
            if (c->opt_branches) da_o = slot(3);   // the two head branches run concurrently: no slot is shared between them
            // convPb fused into the detector-head kernel: convPa.3's output must outlive the network pass, so convDa.3
            // writes over the backbone output instead (slot 0) -- ConvSta, its last reader, then runs before the heads
            if (c->skip_pb_now) da_o = slot(0);
        }
    }
    const DevPtr *x = &a3b;
    static const char *nm1[3] = {"conv4.0.conv1", "conv4.1.conv1", "conv4.2.conv1"};
    static const char *nm2[3] = {"conv4.0.conv2", "conv4.1.conv2", "conv4.2.conv2"};
    static const char *nm3[3] = {"conv4.0.conv3", "conv4.1.conv3", "conv4.2.conv3"};
    // one ResBlock (nets/sfd2.py:25-55) in plain fp16: the fused kernel wherever the fused path runs (SFD2_FUSED_RB=0 in
    // experiment builds: three kernels per block)
    const char *frb = sfd2_env("SFD2_FUSED_RB");
    const bool fused_rb = c->fuse_now != 0 && !(frb && frb[0] == '0');
    static const char *nmf[3] = {"conv4.0", "conv4.1", "conv4.2"};
    auto rb_f16 = [&](int b) {
        DevPtr &t1 = t1v[b], &t2 = t2v[b], &ob = rov[b];
        if (fused_rb && c->rb1[b].wrm.p && c->rb3[b].wrm.p) {
            ProfScope ps(c, nmf[b], "resblock_kernel", 2.0 * P4 * 256 * (256 + 72 + 256), P4 * 256 * 4);
            launch_resblock(st, x->as<half_t>(), H4, W4, c->rb1[b].wrm.as<half_t>(), c->rb1[b].scale.as<float>(),
                            c->rb1[b].shift.as<float>(), c->rb2[b].wgc.as<half_t>(), c->rb2[b].scale.as<float>(),
                            c->rb2[b].shift.as<float>(), c->rb3[b].wrm.as<half_t>(), c->rb3[b].scale.as<float>(),
                            c->rb3[b].shift.as<float>(), alias ? t1.as<half_t>() : ob.as<half_t>(), c->zero_page.as<half_t>());
            x = alias ? &t1 : &ob;   // the fused kernel must not write over its own input: with the arena the output takes t1's slot
            return;
        }
        conv(c, nm1[b], c->rb1[b], *x, H4, W4, t1, H4, W4, 1);
        {
            ProfScope ps(c, nm2[b], "gconv3x3_g8_kernel", 2.0 * P4 * 256 * 72, P4 * 256 * 4);
            launch_gconv3x3_g8(st, t1.as<half_t>(), H4, W4, c->rb2[b].w.as<half_t>(),
                               c->rb2[b].scale.as<float>(), c->rb2[b].shift.as<float>(), t2.as<half_t>());
        }
        conv(c, nm3[b], c->rb3[b], t2, H4, W4, ob, H4, W4, 1, x->as<half_t>());
        x = &ob;
    };
    if (comp) {
        // SFD2_PREC_F16C backbone: every activation carries a corr plane, every layer adds the fp8 correction terms
        // Option "fp6_acts": three tensors have conv3x3_pp<comp> as their only reader -- conv1b's output (-> conv2a), conv2b's (-> conv3a) and
        // conv3a's (-> conv3b).  Their corr records leave the producer as block-scaled fp6 half-records, and the consumer's correction is one
        // fp6 x fp6 scaled MFMA per unit: 33.5 cycles where a unit with fp8 on either side takes 66 (profiles/r04_mfma_probe.txt).  Decided
        // per tensor: the producer must be the kernel that can write them.
        const bool use6 = c->opt_fp6_acts && !c->opt_generic_c && c->c2a.wc66.p && c->c3a.wc66.p && c->c3b.wc66.p;
        const bool s6 = use6 && c->fuse_now && c->w1b_stem_c6.p;                                                                   // a1b (the fused stem writes it)
        const bool b6 = use6 && !c->opt_no_rf_c && conv3x3_rf_c_serves(3, 2, c->c2b.cout_pad, c->c2b.cin, H4, W4);   // a2b (conv3x3_rf<2,comp>)
        const bool a6 = use6;                                                                                   // a3a (conv3x3_pp<comp>)
        // Option "s2d" (throughput path only: the tensor is not readable by sfd2_debug_activation in that layout): conv2a stores its output as
        // four parity planes at quarter resolution, conv2b reads them as a stride-1 layer (conv2b_s2d_kernel.hip)
        const bool d2 = c->opt_s2d && alias && s6 && !c->opt_no_rf_c && conv2b_s2d_serves(H2, W2, c->c2b.cin, c->c2b.cout_pad) &&
                        H4 * 2 == H2 && W4 * 2 == W2;
        {
            static const char *t6[3] = {"bn1b", "bn2b", "conv3a"};
            const bool f6[3] = {s6, b6, a6};
            for (int i = 0; i < 3; ++i) {
                auto it = c->acts.find(t6[i]);
                if (it != c->acts.end()) it->second.fmt6 = f6[i];
            }
        }
        if (c->fuse_now && !c->opt_generic_c) {
            ProfScope ps(c, "conv1a+conv1b", "fused_stem_c_kernel", 2.0 * P1 * 64 * 27 + 2.0 * (double)H2 * W2 * 64 * 576,
                         P1 * 12 + (double)H2 * W2 * 256);
            launch_fused_stem_c(st, img_dev, H, W, normalise, c->c1a.wc.as<half_t>(), c->c1a.scale.as<float>(),
                                c->c1a.shift.as<float>(), s6 ? c->w1b_stem_c6.p : c->w1b_stem_c.p, c->c1b.scale.as<float>(), c->c1b.shift.as<float>(),
                                a1b.as<half_t>(), corr_of(a1b, (size_t)H2 * W2, 64), H2, W2, c->c1b.sbyte, c->range_stat.as<unsigned int>(), s6 ? 2 : 0);
        } else {
            {
                ProfScope ps(c, "conv1a", "conv1a_c_kernel", 2.0 * P1 * 64 * 27, P1 * (12 + 256));
                launch_conv1a_c(st, img_dev, H, W, normalise, c->c1a.wc.as<half_t>(), c->c1a.scale.as<float>(),
                                c->c1a.shift.as<float>(), c->a1a.as<half_t>(), corr_of(c->a1a, (size_t)H * W, 64), range_slot(c, SFD2_RS_CONV1A));
            }
            convc(c, "conv1b", c->c1b, c->a1a, H, W, a1b, H2, W2, 1, true, true, nullptr, SFD2_RS_CONV1B);
        }
        convc(c, "conv2a", c->c2a, a1b, H2, W2, a2a, H2, W2, 1, true, true, nullptr, SFD2_RS_CONV2A, (s6 ? 1 : 0) | (d2 ? 4 : 0));
        if (d2) {
            const ConvW &L = c->c2b;
            ProfScope ps(c, "conv2b", "conv2b_s2d_kernel", 2.0 * P4 * L.cout * L.cin * 9, 4.0 * ((double)H2 * W2 * L.cin + (double)L.cout * L.cin * 9) + P4 * L.cout_pad * 4.0);
            launch_conv2b_s2d(st, a2a.as<half_t>(), corr_of(a2a, (size_t)H2 * W2, L.cin), H4, W4, L.wc.as<half_t>(), L.scale.as<float>(), L.shift.as<float>(), 1,
                              a2b.as<half_t>(), corr_of(a2b, (size_t)H4 * W4, L.cout_pad), c->zero_page.as<half_t>(), L.sbyte, range_slot(c, SFD2_RS_CONV2B), b6 ? 2 : 0);
        } else
        convc(c, "conv2b", c->c2b, a2a, H2, W2, a2b, H4, W4, 1, true, true, nullptr, SFD2_RS_CONV2B, b6 ? 2 : 0);
        // Option "c3b_plain" (the load-time self-check decides, api_weights.hip): conv3b takes the hi plane of conv3a's output only; conv3a keeps writing
        // its records unless SFD2_C3A_PLAIN_OUT says otherwise (experiment switch until measured)
        const bool p3b = c->opt_c3b_plain && a6 && b6 && !c->opt_generic_c;
        static const bool c3a_plain_out = sfd2_env("SFD2_C3A_KEEP_CORR") == nullptr;
        const bool p3a = p3b && c3a_plain_out;
        {
            auto it = c->acts.find("conv3a");
            if (it != c->acts.end() && it->second.p == a3a.p) {      // (sfd2_debug_activation: the tensor has no corr plane then)
                it->second.pc = p3a ? nullptr : corr_of(a3a, (size_t)H4 * W4, 256);
                it->second.fmt6 = a6 && !p3a;
            }
        }
        convc(c, "conv3a", c->c3a, a2b, H4, W4, a3a, H4, W4, 1, true, !p3a, nullptr, SFD2_RS_CONV3A, (b6 ? 1 : 0) | ((a6 && !p3a) ? 2 : 0));
        // Option "trunk_r1": the three tensors the ResBlocks read as block input (conv3b's output, the outputs of blocks 0 and 1) with the residual
        // byte only -- their readers are conv1x1_c256_c<.., 2, ..> (value bytes rebuilt from the hi plane) and rb23_c_kernel's skip path
        const bool tr1 = c->opt_trunk_r1 && a6 && !c->opt_generic_c && c->opt_comp_rb && c->opt_rb_inner >= 2 && c->opt_fuse_rb23 &&
                         c->rb1[0].wfl.p && c->rb3[0].wfl.p && c->rb2[0].wlk.p && c->rb1[0].wfr.p && c->rb3[0].wf8l.p;
        {
            static const char *tr[3] = {"bn3b", "conv4.0", "conv4.1"};
            for (int i = 0; i < 3; ++i) {
                auto it = c->acts.find(tr[i]);
                if (it != c->acts.end()) it->second.r1 = tr1;
            }
        }
        convc(c, "conv3b", c->c3b, a3a, H4, W4, a3b, H4, W4, 1, !p3b, true, nullptr, SFD2_RS_CONV3B, ((a6 && !p3b) ? 1 : 0) | (tr1 ? 8 : 0));
        for (int b = 0; b < 3; ++b) {  // ResBlock (nets/sfd2.py:25-55)
            if (!c->opt_comp_rb) { rb_f16(b); continue; }   // option "comp_rb" = 0: this block in plain fp16 on the hi planes
            DevPtr &t1 = t1v[b], &t2 = t2v[b], &ob = rov[b];
            const int inner = (!c->opt_generic_c && c->rb1[b].wfl.p && c->rb3[b].wfl.p && c->rb2[b].wlk.p) ? c->opt_rb_inner : 0;
            if (inner) {
                // Option "rb_inner": the tensors INSIDE the block as plain fp16 (1: t2, 2: t1 and t2).  These kernels are bound
                // by HBM bytes, a plain tensor is half of a compensated one; the filters stay compensated (over a plain input the
                // residual term x * lo_w is a second fp16 pass: there is no fp8 value byte of x to feed the scaled MFMA).
                const size_t PP = (size_t)H4 * W4;
                const ConvW &L1 = c->rb1[b], &L2 = c->rb2[b], &L3 = c->rb3[b];
                const bool t1p = inner >= 2;
                {   // sfd2_debug_activation: these tensors have no corr plane on this path
                    auto i1 = c->acts.find(std::string(nm1[b]).substr(0, 8) + "bn1"), i2 = c->acts.find(std::string(nm1[b]).substr(0, 8) + "bn2");
                    if (i1 != c->acts.end() && i1->second.p == t1.p) i1->second.pc = t1p ? nullptr : corr_of(t1, PP, 256);
                    if (i2 != c->acts.end() && i2->second.p == t2.p) { i2->second.pc = nullptr; i2->second.absent = t1p && c->opt_fuse_rb23; }
                }
                if (t1p) {
                    ProfScope ps(c, nm1[b], "conv1x1_c256<comp,plain out>", 2.0 * P4 * 256.0 * 256.0, P4 * 256.0 * (tr1 ? 5 : 6));
                    launch_conv1x1_c256_c(st, x->as<half_t>(), corr_of(*x, PP, 256), (int)PP, L1.wfh.as<half_t>(), tr1 ? L1.wfr.as<half_t>() : L1.wfc.as<half_t>(),
                                          L1.scale.as<float>(), L1.shift.as<float>(), 1, nullptr, nullptr, t1.as<half_t>(), nullptr,
                                          c->zero_page.as<half_t>(), L1.sbyte, range_slot(c, SFD2_RS_T1_0 + b), tr1 ? 1 : 0);
                } else {
                    convc(c, nm1[b], L1, *x, H4, W4, t1, H4, W4, 1, true, true, nullptr, SFD2_RS_T1_0 + b);
                }
                if (t1p && c->opt_fuse_rb23) {
                    // (the last block's output keeps its units when a compensated head layer will read them)
                    const bool out_r1 = b < 2 || !(c->opt_comp_heads || c->opt_comp_det);
                    const int r1f = tr1 ? (1 | (out_r1 ? 2 : 0)) : 0;
                    if (b == 2) { auto it = c->acts.find("conv4.2"); if (it != c->acts.end()) it->second.r1 = tr1 && out_r1; }
                    ProfScope ps(c, nm3[b], "rb23_c_kernel", 2.0 * P4 * 256 * 72 + 2.0 * P4 * 256.0 * 256.0, P4 * 256.0 * (10 - (r1f & 1) - ((r1f >> 1) & 1)));
                    launch_rb23_c(st, t1.as<half_t>(), H4, W4, L2.w.as<half_t>(), L2.wlk.as<half_t>(), L2.scale.as<float>(), L2.shift.as<float>(),
                                  L3.wfh.as<half_t>(), L3.wfl.as<half_t>(), L3.scale.as<float>(), L3.shift.as<float>(), x->as<half_t>(),
                                  corr_of(*x, PP, 256), ob.as<half_t>(), corr_of(ob, PP, 256), c->zero_page.as<half_t>(),
                                  range_slot(c, SFD2_RS_T2_0 + b), range_slot(c, SFD2_RS_OUT_0 + b), r1f, L3.wf8l.as<half_t>(), L3.sbyte);
                    x = &ob;
                    continue;
                }
                {
                    ProfScope ps(c, nm2[b], t1p ? "gconv_c_kernel<plain>" : "gconv_c_kernel<plain out>", 2.0 * P4 * 256 * 72, P4 * 256 * (t1p ? 4 : 6));
                    launch_gconv_c(st, t1.as<half_t>(), t1p ? nullptr : corr_of(t1, PP, 256), H4, W4, L2.w.as<half_t>(),
                                   t1p ? L2.wlk.p : L2.wc.p, L2.scale.as<float>(), L2.shift.as<float>(), t2.as<half_t>(), nullptr, L2.sbyte, 0, H4,
                                   range_slot(c, SFD2_RS_T2_0 + b));
                }
                {
                    ProfScope ps(c, nm3[b], "conv1x1_c256<comp,plain in>+res", 2.0 * P4 * 256.0 * 256.0, P4 * 256.0 * 10);
                    launch_conv1x1_c256_c(st, t2.as<half_t>(), nullptr, (int)PP, L3.wfh.as<half_t>(), L3.wfl.as<half_t>(), L3.scale.as<float>(),
                                          L3.shift.as<float>(), 1, x->as<half_t>(), corr_of(*x, PP, 256), ob.as<half_t>(), corr_of(ob, PP, 256),
                                          c->zero_page.as<half_t>(), L3.sbyte, range_slot(c, SFD2_RS_OUT_0 + b));
                }
            } else {
                {
                    auto i1 = c->acts.find(std::string(nm1[b]).substr(0, 8) + "bn1"), i2 = c->acts.find(std::string(nm1[b]).substr(0, 8) + "bn2");
                    if (i1 != c->acts.end() && i1->second.p == t1.p) i1->second.pc = corr_of(t1, (size_t)H4 * W4, 256);
                    if (i2 != c->acts.end() && i2->second.p == t2.p) { i2->second.pc = corr_of(t2, (size_t)H4 * W4, 256); i2->second.absent = false; }
                }
                convc(c, nm1[b], c->rb1[b], *x, H4, W4, t1, H4, W4, 1, true, true, nullptr, SFD2_RS_T1_0 + b);
                {
                    ProfScope ps(c, nm2[b], "gconv_c_kernel", 2.0 * P4 * 256 * 72, P4 * 256 * 8);
                    launch_gconv_c(st, t1.as<half_t>(), corr_of(t1, (size_t)H4 * W4, 256), H4, W4, c->rb2[b].w.as<half_t>(),
                                   c->rb2[b].wc.p, c->rb2[b].scale.as<float>(), c->rb2[b].shift.as<float>(), t2.as<half_t>(),
                                   corr_of(t2, (size_t)H4 * W4, 256), c->rb2[b].sbyte, 0, H4, range_slot(c, SFD2_RS_T2_0 + b));
                }
                convc(c, nm3[b], c->rb3[b], t2, H4, W4, ob, H4, W4, 1, true, true, x, SFD2_RS_OUT_0 + b);
            }
            x = &ob;
        }
    } else {
    if (c->fuse_now) {
        ProfScope ps(c, "conv1a+conv1b", "fused_stem_kernel", 2.0 * P1 * 64 * 27 + 2.0 * (double)H2 * W2 * 64 * 576,
                     P1 * 12 + (double)H2 * W2 * 128);
        launch_fused_stem(st, img_dev, H, W, normalise, c->c1a.w.as<half_t>(), c->c1a.scale.as<float>(),
                          c->c1a.shift.as<float>(), c->w1b_fused.as<half_t>(), c->c1b.scale.as<float>(),
                          c->c1b.shift.as<float>(), a1b.as<half_t>(), H2, W2);
    } else {
        {
            ProfScope ps(c, "conv1a", "conv1a_kernel", 2.0 * P1 * 64 * 27, P1 * (12 + 128));
            launch_conv1a(st, img_dev, H, W, normalise, c->c1a.w.as<half_t>(), c->c1a.scale.as<float>(),
                          c->c1a.shift.as<float>(), c->a1a.as<half_t>());
        }
        conv(c, "conv1b", c->c1b, c->a1a, H, W, a1b, H2, W2, 1);
    }
    conv(c, "conv2a", c->c2a, a1b, H2, W2, a2a, H2, W2, 1);
    conv(c, "conv2b", c->c2b, a2a, H2, W2, a2b, H4, W4, 1);
    conv(c, "conv3a", c->c3a, a2b, H4, W4, a3a, H4, W4, 1);
    conv(c, "conv3b", c->c3b, a3a, H4, W4, a3b, H4, W4, 1);
    for (int b = 0; b < 3; ++b) rb_f16(b);
    }
    // The two head branches read the backbone output and nothing of each other (nets/sfd2.py:328-342).  The detector
    // branch works on the 1/8 map (convPa.3: 133 tiles for 256 CUs at 1600x1200), so on its own it leaves part of the
    // chip idle; with option "branches" it runs on a side stream beside the descriptor branch (fork / join by events;
    // inside a captured hipGraph this becomes a fork in the graph).  Measured (tools/ab_branches.py, interleaved A/B at
    // 1600x1200): 1.260 -> 1.238 ms per extract (-1.7 %), outputs bit-identical.  Off by default: overlapped launches
    // stretch each other's event-timed durations, and bench.py's per-kernel roofline wants uncontended ones.
    const bool sta_early = alias && c->skip_pb_now;
    // Option "sta_side" (round 6): ConvSta -- 63 MB read at HBM speed, 16 us -- on the side stream beside convPa.0 / convPa.3 / convDa.0 (matrix-bound, another part of
    // the chip's budget); fork here, join at the end of this function (the heat-map kernel behind it is the first reader of its output).  Inside a captured hipGraph: a
    // fork in the graph.  Same kernel, same bits.  Not with "branches" (that option owns the side stream).
    const bool sta_side = c->has_sta && sta_early && c->opt_sta_side && !c->opt_branches;
    if (c->has_sta && sta_early) {
        if (sta_side) {
            HIPCHECK(hipEventRecord(c->ev_fork, st));
            HIPCHECK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
            c->cur_stream = c->side_stream;
        }
        {
            ProfScope ps(c, "ConvSta", "convsta_kernel", 2.0 * P4 * 3 * 256, P4 * (512 + 12));
            launch_convsta(c->cur_stream, x->as<half_t>(), H4 * W4, c->sta_w16.as<float>(), c->sta_b.as<float>(), c->sta.as<float>());
        }
        if (sta_side) {
            HIPCHECK(hipEventRecord(c->ev_join, c->side_stream));
            c->cur_stream = st;
        }
    }
    const bool fork = c->opt_branches != 0;
    if (fork) {
        HIPCHECK(hipEventRecord(c->ev_fork, st));
        HIPCHECK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        c->cur_stream = c->side_stream;
    }
    // option "comp_heads": the four 3x3 layers of the head branches compensated as well (their inputs then need corr planes: the
    // backbone output has one when the ResBlocks are compensated); convPb / convDb / ConvSta read hi planes either way
    // option "comp_det": the detector branch only (convPa.0, convPa.3) -- its score goes through exp(), which is what moves key points
    const bool ch = comp && c->opt_comp_heads && c->opt_comp_rb;
    const bool chp = ch || (comp && c->opt_comp_det && c->opt_comp_rb);
    if (chp) {
        convc(c, "convPa.0", c->pa0, *x, H4, W4, pa0_o, H8, W8, 1, true, true, nullptr, SFD2_RS_PA0);
        convc(c, "convPa.3", c->pa3, pa0_o, H8, W8, pa_o, H8, W8, 0, true, false);
    } else {
        conv(c, "convPa.0", c->pa0, *x, H4, W4, pa0_o, H8, W8, 1);
        conv(c, "convPa.3", c->pa3, pa0_o, H8, W8, pa_o, H8, W8, 0);
    }
    c->pa_cur = pa_o.as<half_t>();
    if (!c->skip_pb_now) conv(c, "convPb", c->pb, pa_o, H8, W8, c->logits, H8, W8, 0, nullptr, 1);
    if (!c->skip_head_now) {
        ProfScope ps(c, "detector_head", "detector_head_kernel", 0.0, P8 * (65 * 4 + 256));
        launch_detector_head(c->cur_stream, c->logits.as<float>(), 128, H8, W8, c->score.as<float>());
    }
    if (fork) {
        HIPCHECK(hipEventRecord(c->ev_join, c->side_stream));
        c->cur_stream = st;
    }
    if (ch) {
        convc(c, "convDa.0", c->da0, *x, H4, W4, da0_o, H4, W4, 1, true, true, nullptr, SFD2_RS_DA0);
        convc(c, "convDa.3", c->da3, da0_o, H4, W4, da_o, H4, W4, 0, true, false);
    } else {
        conv(c, "convDa.0", c->da0, *x, H4, W4, da0_o, H4, W4, 1);
        if (!c->skip_da3_now) conv(c, "convDa.3", c->da3, da0_o, H4, W4, da_o, H4, W4, 0);
    }
    c->da_cur = da_o.as<half_t>();
    c->da0_cur = da0_o.as<half_t>();
    if (!c->skip_db_now) conv(c, "convDb", c->db, da_o, H4, W4, c->draw, H4, W4, 0, nullptr, 1);
    if (c->has_sta && !sta_early) {
        ProfScope ps(c, "ConvSta", "convsta_kernel", 2.0 * P4 * 3 * 256, P4 * (512 + 12));
        launch_convsta(st, x->as<half_t>(), H4 * W4, c->sta_w16.as<float>(), c->sta_b.as<float>(), c->sta.as<float>());
    }
    if (fork || sta_side) HIPCHECK(hipStreamWaitEvent(st, c->ev_join, 0));
    HIPCHECK(hipGetLastError());
    if (c->net_error) { c->net_error = 0; return -1; }      // (a layer helper recorded an error: sfd2_last_error has its text)
    return 0;
}
