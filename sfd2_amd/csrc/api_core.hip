// libsfd2hip: context life cycle, options, profiling and timings of the C-ABI declared in include/sfd2_hip.h.
#include "sfd2_ctx.h"

static thread_local std::string g_err;
int sfd2_fail(const std::string &m)
{
    g_err = m;
    return -1;
}

std::atomic<unsigned long long> g_alloc_gen{0};

thread_local int g_sfd2_cu_limit = 0;       // set per network pass from the context's option "cu_limit" (sfd2_internal.h)

// ------------------------------------------------------------------------------------------ basics
extern "C" int sfd2_version(void) { return 108; }   // 108: + sfd2_get_option; 107: + sfd2_get_relax_status (option c3b_plain); 105: + sfd2_extract_record_async, sfd2_desc_pack, SFD2_FLAG_ASYNC with host outputs (round 5); 106: + sfd2_get_margin_status
extern "C" const char *sfd2_last_error(void) { return g_err.c_str(); }

extern "C" int sfd2_ctx_create(int device, sfd2_ctx **out)
{
    if (!out) return fail("sfd2_ctx_create: out is null");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail("sfd2_ctx_create: no HIP device available (libsfd2hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail("sfd2_ctx_create: bad device index");
    HIPCHECK(hipSetDevice(device));
    sfd2_ctx *c = new sfd2_ctx();
    c->device = device;
    HIPCHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; ++i) HIPCHECK(hipEventCreate(&c->ev[i]));
    HIPCHECK(hipEventCreateWithFlags(&c->ev_jobs, hipEventDisableTiming));
    HIPCHECK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIPCHECK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    c->cur_stream = c->stream;
    for (int i = 0; i < SFD2_IMG_SLOTS; ++i) {
        HIPCHECK(hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&c->ev_img_free[i], hipEventDisableTiming));
    }
    c->fuse = sfd2_env("SFD2_NO_FUSE") ? 0 : 1;
    // the zero page, and behind it the range-status words (conv3x3_pp reaches them through the zero-page pointer it holds anyway)
    // ... and behind those the device-side history of folded maxima (SFD2_RS_COUNT words) and the record of the last asynchronous extract (4 words)
    HIPCHECK(c->zero_page.ensure(SFD2_ZERO_PAGE_BYTES + ((size_t)SFD2_RS_COUNT * (SFD2_RANGE_SUB + 1) + 4) * sizeof(unsigned int)));
    HIPCHECK(hipMemset(c->zero_page.p, 0, c->zero_page.cap));
    c->range_stat.p = c->zero_page.as<char>() + SFD2_ZERO_PAGE_BYTES;
    c->range_hist_dev.p = c->range_stat.as<unsigned int>() + SFD2_RS_COUNT * SFD2_RANGE_SUB;
    c->extract_rec.p = c->range_hist_dev.as<unsigned int>() + SFD2_RS_COUNT;
    static_assert(SFD2_RS_COUNT == SFD2_RANGE_TENSORS && AE_COUNT == SFD2_RANGE_GROUPS, "include/sfd2_hip.h and the internal tables agree");

    *out = c;
    return 0;
}

// Where a device sits on the host (version 108): its PCI address "dddd:bb:dd.f" (hipDeviceGetPCIBusId), *n_devices = HIP's device count.  What a one-process-
// per-GPU launcher needs to keep a rank's decoder / writer threads on the socket its GPU hangs off (sfd2_amd/sharding.py pin_to_gpu_socket reads
// /sys/bus/pci/devices/<address>/local_cpulist); the reference leaves placement to the DataLoader's worker processes (extract_localization.py:230-233).
extern "C" int sfd2_device_pci_bus_id(int device, char *out, int len, int *n_devices)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail("sfd2_device_pci_bus_id: no HIP device available");
    if (n_devices) *n_devices = ndev;
    if (device < 0 || device >= ndev) return fail("sfd2_device_pci_bus_id: bad device index");
    if (!out || len < 13) return fail("sfd2_device_pci_bus_id: buffer of at least 13 bytes needed");
    HIPCHECK(hipDeviceGetPCIBusId(out, len, device));
    return 0;
}

extern "C" void sfd2_ctx_destroy(sfd2_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    graphs_release(c);
    for (int i = 0; i < 4; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_jobs) (void)hipEventDestroy(c->ev_jobs);
    for (int i = 0; i < SFD2_IMG_SLOTS; ++i) {
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_img_free[i]) (void)hipEventDestroy(c->ev_img_free[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
    if (c->pin_jobs) (void)hipHostFree(c->pin_jobs);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;      // every DevBuf / ConvW member frees its own allocation (sfd2_ctx.h)
}

extern "C" void *sfd2_get_stream(sfd2_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int sfd2_set_precision(sfd2_ctx *c, int mode)
{
    if (!c) return fail("sfd2_set_precision: null ctx");
    if (mode != SFD2_PREC_F16 && mode != SFD2_PREC_F32 && mode != SFD2_PREC_F16X3 && mode != SFD2_PREC_F16C)
        return fail("sfd2_set_precision: unknown mode");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (mode != c->precision) graphs_release(c);   // a captured unit holds the kernels of the precision it was captured in
    c->precision = mode;
    return sfd2_margin_selfcheck_if_pending(c);     // (a context that was loaded in another precision: the F16C self-check runs now)
}

extern "C" int sfd2_set_option(sfd2_ctx *c, const char *key, int value)
{
    if (!c || !key) return fail("sfd2_set_option: null argument");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    const std::string k(key);
    graphs_release(c);   // every option below decides which kernels a captured unit contains
    if (k == "fuse") c->fuse = value ? 1 : 0;
    else if (k == "fuse_det") c->fuse_det = value ? 1 : 0;
    else if (k == "alias") c->opt_alias = value ? 1 : 0;
    else if (k == "graphs") c->use_graphs = value ? 1 : 0;
    else if (k == "branches") c->opt_branches = value ? 1 : 0;
    else if (k == "sta_side") c->opt_sta_side = value ? 1 : 0;
    else if (k == "fuse_post") c->opt_fuse_post = value ? 1 : 0;
    else if (k == "sparse_desc") c->opt_sparse_desc = value ? 1 : 0;
    else if (k == "sparse_da3") c->opt_sparse_da3 = value ? 1 : 0;
    else if (k == "cu_limit") c->opt_cu_limit = value < 0 ? 0 : value;
    else if (k == "auto_range") c->opt_auto_range = value ? 1 : 0;
    else if (k == "range_fallback") c->opt_range_fallback = value ? 1 : 0;
    else if (k == "x3_pp") c->opt_x3_pp = value ? 1 : 0;
    else if (k == "x3_desc16") c->opt_x3_desc16 = value ? 1 : 0;
    else if (k == "auto_margin") c->opt_auto_margin = value ? 1 : 0;
    else if (k == "fp6_filters") c->opt_fp6_filters = value ? 1 : 0;
    else if (k == "fp6_acts") c->opt_fp6_acts = value ? 1 : 0;
    else if (k == "s2d") c->opt_s2d = value ? 1 : 0;
    else if (k == "trunk_r1") c->opt_trunk_r1 = value ? 1 : 0;
    else if (k == "fuse_pb") c->opt_fuse_pb = value ? 1 : 0;
    else if (k == "generic_c") c->opt_generic_c = value ? 1 : 0;
    else if (k == "comp_rb") c->opt_comp_rb = value ? 1 : 0;
    else if (k == "no_rf_c") c->opt_no_rf_c = value ? 1 : 0;
    else if (k == "comp_heads") { c->opt_comp_heads = c->user_comp_heads = value ? 1 : 0; c->user_set_comp_heads = true; }
    else if (k == "c3b_plain") { c->user_c3b_plain = value < 0 ? -1 : (value ? 1 : 0); c->opt_c3b_plain = value > 0 ? 1 : 0; }
    else if (k == "comp_det") c->opt_comp_det = value ? 1 : 0;
    else if (k == "fuse_rb23") c->opt_fuse_rb23 = value ? 1 : 0;
    else if (k == "rb_inner") { c->opt_rb_inner = c->user_rb_inner = value < 0 ? 0 : (value > 2 ? 2 : value); c->user_set_rb_inner = true; }
    else return fail("sfd2_set_option: unknown key '" + k + "'");
    return 0;
}

// What the context RUNS with for a key of sfd2_set_option (version 108): the value last set, or -- for the keys the load-time self-check of
// SFD2_PREC_F16C decides ("rb_inner", "comp_heads", "c3b_plain") -- its choice.  A second context that is to compute the same bits as this one copies
// these instead of replaying what the caller asked for (sfd2_amd/model.py replica()).
extern "C" int sfd2_get_option(sfd2_ctx *c, const char *key, int *value)
{
    if (!c || !key || !value) return fail("sfd2_get_option: null argument");
    const std::string k(key);
    const struct { const char *name; int v; } tab[] = {
        {"fuse", c->fuse}, {"fuse_det", c->fuse_det}, {"alias", c->opt_alias}, {"graphs", c->use_graphs}, {"branches", c->opt_branches}, {"sta_side", c->opt_sta_side},
        {"fuse_post", c->opt_fuse_post}, {"sparse_desc", c->opt_sparse_desc}, {"sparse_da3", c->opt_sparse_da3}, {"cu_limit", c->opt_cu_limit},
        {"auto_range", c->opt_auto_range}, {"range_fallback", c->opt_range_fallback}, {"x3_pp", c->opt_x3_pp}, {"x3_desc16", c->opt_x3_desc16},
        {"auto_margin", c->opt_auto_margin}, {"fp6_filters", c->opt_fp6_filters}, {"fp6_acts", c->opt_fp6_acts}, {"s2d", c->opt_s2d},
        {"trunk_r1", c->opt_trunk_r1}, {"fuse_pb", c->opt_fuse_pb}, {"generic_c", c->opt_generic_c}, {"comp_rb", c->opt_comp_rb},
        {"no_rf_c", c->opt_no_rf_c}, {"comp_heads", c->opt_comp_heads}, {"c3b_plain", c->opt_c3b_plain}, {"comp_det", c->opt_comp_det},
        {"fuse_rb23", c->opt_fuse_rb23}, {"rb_inner", c->opt_rb_inner},
    };
    for (const auto &e : tab)
        if (k == e.name) { *value = e.v; return 0; }
    return fail("sfd2_get_option: unknown key '" + k + "'");
}

extern "C" int sfd2_sync(sfd2_ctx *c)
{
    if (!c) return fail("sfd2_sync: null ctx");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int sfd2_set_profiling(sfd2_ctx *c, int max_steps)
{
    if (!c) return fail("sfd2_set_profiling: null ctx");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    if (max_steps < 0 || max_steps > 4096) return fail("sfd2_set_profiling: max_steps out of range");
    const size_t need = (size_t)max_steps * PROF_SLOTS * 2;
    while (c->prof_ev.size() < need) {
        hipEvent_t e;
        HIPCHECK(hipEventCreate(&e));
        c->prof_ev.push_back(e);
    }
    c->prof_max_steps = max_steps;
    c->prof_step = 0;
    c->prof_slot = 0;
    c->prof_used.assign(max_steps, 0);
    c->prof_row.assign((size_t)max_steps * PROF_SLOTS, 0);
    c->prof_tab.clear();
    return 0;
}

extern "C" int sfd2_set_profile_filter(sfd2_ctx *c, const char *substr)
{
    if (!c) return fail("sfd2_set_profile_filter: null ctx");
    c->prof_filter = substr ? substr : "";
    return 0;
}

extern "C" int sfd2_get_layer_timings(sfd2_ctx *c, sfd2_layer_timing *out, int cap, int *n)
{
    if (!c || !n) return fail("sfd2_get_layer_timings: null argument");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int st = 0; st < c->prof_step; ++st)
        for (int sl = 0; sl < c->prof_used[st]; ++sl) {
            float ms = 0.0f;
            const size_t e = ((size_t)st * PROF_SLOTS + sl) * 2;
            const int row = c->prof_row[(size_t)st * PROF_SLOTS + sl];
            if (hipEventElapsedTime(&ms, c->prof_ev[e], c->prof_ev[e + 1]) == hipSuccess) {
                c->prof_tab[row].ms_total += ms;
                c->prof_tab[row].launches += 1;
            }
        }
    c->prof_step = 0;  // events consumed; the table keeps accumulating until sfd2_set_profiling resets it
    *n = (int)c->prof_tab.size();
    if (out)
        for (int i = 0; i < *n && i < cap; ++i) out[i] = c->prof_tab[i];
    return 0;
}

extern "C" int sfd2_get_timings(sfd2_ctx *c, sfd2_timings *out)
{
    if (!c || !out) return fail("sfd2_get_timings: null argument");
    *out = c->tim;
    return 0;
}

// ------------------------------------------------------------------------------------------ range status (include/sfd2_hip.h)
static const int kRsGroup[SFD2_RS_COUNT] = {AE_CONV1A, AE_CONV1B, AE_CONV2A, AE_CONV2B, AE_CONV3A, AE_TRUNK, AE_T1_0, AE_T1_1, AE_T1_2,
                                            AE_T2_0, AE_T2_1, AE_T2_2, AE_TRUNK, AE_TRUNK, AE_TRUNK, AE_PA0, AE_DA0};
extern "C" const char *sfd2_range_tensor_name(int i)
{
    static const char *names[SFD2_RS_COUNT] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4.0.t1", "conv4.1.t1", "conv4.2.t1",
                                               "conv4.0.t2", "conv4.1.t2", "conv4.2.t2", "conv4.0", "conv4.1", "conv4.2", "convPa.0", "convDa.0"};
    return (i >= 0 && i < SFD2_RS_COUNT) ? names[i] : "";
}

// the running words and the device-side history behind them in one read: [SFD2_RS_COUNT][SFD2_RANGE_SUB] then [SFD2_RS_COUNT]
static int fetch_range_words(sfd2_ctx *c, unsigned int *raw /* SFD2_RS_COUNT * (SFD2_RANGE_SUB + 1) */, bool clear)
{
    const size_t bytes = (size_t)SFD2_RS_COUNT * (SFD2_RANGE_SUB + 1) * sizeof(unsigned int);
    HIPCHECK(hipMemcpyAsync(raw, c->range_stat.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (clear) HIPCHECK(hipMemsetAsync(c->range_stat.p, 0, bytes, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int read_range_status(sfd2_ctx *c, sfd2_range_status *out, int reset)
{
    unsigned int raw[SFD2_RS_COUNT * (SFD2_RANGE_SUB + 1)];
    if (fetch_range_words(c, raw, reset != 0)) return -1;
    std::memset(out, 0, sizeof(*out));
    out->n_tensors = SFD2_RS_COUNT;
    const unsigned int sat_bits = 0x44E00000u;      // 1792.0f: raw compare, so that Inf / NaN patterns count as saturated too
    for (int t = 0; t < SFD2_RS_COUNT; ++t) {
        unsigned int m = raw[SFD2_RS_COUNT * SFD2_RANGE_SUB + t];             // folded away by sfd2_extract_record_async / a synchronous extract's start
        for (int s = 0; s < SFD2_RANGE_SUB; ++s) m = std::max(m, raw[t * SFD2_RANGE_SUB + s]);
        float f;
        std::memcpy(&f, &m, 4);
        const bool sat = m >= sat_bits || c->range_hist[t] >= SFD2_C_SAT;
        if (!(f >= c->range_hist[t])) f = c->range_hist[t];                   // (a NaN pattern stays visible)
        if (reset) c->range_hist[t] = 0.0f;
        const int e = c->act_exp[kRsGroup[t]];
        out->max_stored[t] = f;
        out->max_value[t] = std::ldexp(f, -e);
        out->exponent[t] = e;
        if (sat) out->saturated |= 1u << t;
        if (f > 0.0f && f < 0.03125f) out->low |= 1u << t;
    }
    out->fallbacks = c->range_fallbacks;
    return 0;
}

// sfd2_calibrate_range / sfd2_set_act_exponents: maxima recorded under the previous exponents say nothing about the new scaling
int reset_range_records(sfd2_ctx *c)
{
    if (!c->range_stat.p) return 0;
    HIPCHECK(hipMemsetAsync(c->range_stat.p, 0, (size_t)SFD2_RS_COUNT * (SFD2_RANGE_SUB + 1) * sizeof(unsigned int), c->stream));
    for (int t = 0; t < SFD2_RS_COUNT; ++t) c->range_hist[t] = 0.0f;
    return 0;
}

// In front of a synchronous extraction that may fall back: whatever earlier (asynchronous) calls left in the running words goes
// into the device-side history, so the decision after this image covers this image only.  No host synchronisation.
int range_fold_before_sync_extract(sfd2_ctx *c)
{
    if (c->precision != SFD2_PREC_F16C || !c->opt_range_fallback || c->in_fallback) return 0;
    launch_extract_record(c->stream, c->range_stat.as<unsigned int>(), c->range_hist_dev.as<unsigned int>(), nullptr, 0, 0, nullptr);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sfd2_get_range_status(sfd2_ctx *c, sfd2_range_status *out, int reset)
{
    if (!c || !out) return fail("sfd2_get_range_status: null argument");
    HIPCHECK(hipSetDevice(c->device));
    return read_range_status(c, out, reset);
}

int range_wants_fallback(sfd2_ctx *c)
{
    if (c->precision != SFD2_PREC_F16C || !c->opt_range_fallback || c->in_fallback) return 0;
    unsigned int raw[SFD2_RS_COUNT * SFD2_RANGE_SUB];
    HIPCHECK(hipMemcpyAsync(raw, c->range_stat.p, sizeof(raw), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    bool sat = false;
    float mx[SFD2_RS_COUNT];
    for (int t = 0; t < SFD2_RS_COUNT; ++t) {
        unsigned int m = 0;
        for (int s = 0; s < SFD2_RANGE_SUB; ++s) m = std::max(m, raw[t * SFD2_RANGE_SUB + s]);
        std::memcpy(&mx[t], &m, 4);
        sat = sat || m >= 0x44E00000u;      // bits of SFD2_C_SAT (1792.0f); also true for Inf / NaN patterns
    }
    if (!sat) return 0;
    for (int t = 0; t < SFD2_RS_COUNT; ++t)
        if (!(c->range_hist[t] >= mx[t])) c->range_hist[t] = mx[t];                                // the report keeps what happened
    HIPCHECK(hipMemsetAsync(c->range_stat.p, 0, sizeof(raw), c->stream));                          // the next image starts clean
    return 1;
}

// After an SFD2_FLAG_ASYNC sfd2_extract (include/sfd2_hip.h).
extern "C" int sfd2_extract_record_async(sfd2_ctx *c, sfd2_extract_record *rec, int rec_on_device)
{
    if (!c || !rec) return fail("sfd2_extract_record_async: null argument");
    if (!c->counters.p) return fail("sfd2_extract_record_async: no extract has run on this context");
    HIPCHECK(hipSetDevice(c->device));
    static_assert(sizeof(sfd2_extract_record) == 16, "four words");
    unsigned int *dev = rec_on_device ? reinterpret_cast<unsigned int *>(rec) : c->extract_rec.as<unsigned int>();
    launch_extract_record(c->stream, c->range_stat.as<unsigned int>(), c->range_hist_dev.as<unsigned int>(), c->counters.as<unsigned int>(),
                          c->last_sel_cap, c->cand_cap, dev);
    HIPCHECK(hipGetLastError());
    if (!rec_on_device) HIPCHECK(hipMemcpyAsync(rec, dev, sizeof(sfd2_extract_record), hipMemcpyDeviceToHost, c->stream));
    return 0;
}
