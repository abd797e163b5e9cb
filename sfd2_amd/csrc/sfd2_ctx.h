// libsfd2hip, host side: the context and what the api_*.hip translation units share.  Host C++ only (no kernels).
//   api_core.hip     context life cycle, options, profiling, timings, status
//   api_weights.hip  BatchNorm folding, filter packing for every kernel family, sfd2_load_weights
//   api_network.hip  workspace, per-layer kernel dispatch, the network passes of the four precisions
//   api_extract.hip  sfd2_det / sfd2_extract / pyramids / spp variants / stage entry points
//   api_match.hip    the matcher entry points
//   api_graph.hip    sfd2_extract_match and its hipGraph cache
#pragma once
#include "../../include/sfd2_hip.h"
#include "sfd2_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

int sfd2_fail(const std::string &m);          // sets the thread's last-error text, returns -1
#define fail sfd2_fail
#define HIPCHECK(expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)

extern std::atomic<unsigned long long> g_alloc_gen;   // (process-wide, contexts may live on different threads) bumped whenever a workspace buffer is (re)allocated: captured graphs hold raw pointers

// A device pointer with its capacity.  DevPtr is a non-owning view (arena slots, per-call aliases of the context's buffers);
// DevBuf owns its allocation and frees it when it goes away, so tearing a context or a layer down needs no list of members
// (the hand-kept list had missed six of them).
struct DevPtr {
    void *p = nullptr;
    size_t cap = 0;
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};
struct DevBuf : DevPtr {
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept { p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        ++g_alloc_gen;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct ConvW {                 // one folded + packed layer
    int cin = 0, cout = 0, cout_pad = 0, ks = 0, stride = 1;
    DevBuf w, scale, shift;    // fp16 packed filters, fp32 [cout_pad]
    DevBuf scale_rawin;        // convDa.0: scale for an input at the network's own scale (2^e_out only; option "x3_desc16" feeds it f16x3's hi plane)
    DevBuf wsl;                // convDa.3: w in sparse_da3_kernel's fragment order -- [chunk][tap][cout_pad / 32][2][64 lanes][8]: a fragment load reads one contiguous kilobyte
    DevBuf wrm;                // 1x1 256 -> 256 layers: the same filters as plain [cout][cin] fp16 (conv1x1_c256_kernel)
    DevBuf wfh, wfc, wfl;      // conv1x1_c256_c_kernel: the same filters, their corr units and fp16 of (w - fp16(w)) * 2^11 (the second
                               // fp16 pass over a PLAIN input, option "rb_inner") in the kernel's fragment order [8 waves][8][64 lanes][16]
    DevBuf wlk;                // grouped 3x3: the same residuals in w's fragment layout (gconv_c_kernel<false, false>)
    DevBuf wgc;                // grouped 3x3: compact [256 oc][9 taps][8 in] fp16 (resblock_kernel)
    DevBuf wx3;                // fp32 layers, SFD2_PREC_F16X3: every float4 of w as (4 hi, 4 lo) fp16, made on first use
    DevBuf wx3p;               // ... or as two planes (hi, lo') for conv3x3_pp's three-pass instantiation (3x3 stride-1 layers)
    DevBuf wc;                 // SFD2_PREC_F16C: [2 * cin / 32][taps][cout_pad][32] units -- the fp16 filters in 32-wide chunks, then the
                               // corr units (fp8 of w * 2^b0, fp8 of (w - fp16(w)) * 2^(b0 + 11)); conv1a / grouped conv: hi then lo fragments
    int sbyte = 127;           // E8M0 scale byte of the layer's corr MFMAs: 127 - 9 - b0
    DevBuf wf8l;               // the 1x1 layers' filter residuals as e4m3 in rb23_c_kernel's K order (option "trunk_r1": conv3's x * lo_w term on the scaled MFMA)
    DevBuf wfr;                // the 1x1 layers' corr units in the order of the residual-only input form (option "trunk_r1")
    DevBuf wc6;                // conv3x3_pp layers: wc with the corr filter rows as fp6 (e2m3) strings, and ...
    DevBuf sa6;                // ... [shift[cout_pad] | per-output-channel E8M0 scale bytes, replicated into the four bytes of an int, [cout_pad]]
    size_t w_floats = 0;       // floats in w (fp32 layers)
    std::vector<float> h_scale, h_shift;   // host copies of the folded constants as loaded (before the activation exponents)
    std::vector<int> h_sa6;                // host copy of sa6
    DevBuf wc66, sa66;                     // the same two arrays for fp6 PIXEL records (option "fp6_acts"): the strings in the half-records' channel
    std::vector<int> h_sa66;               // order (sfd2_epi16_fp6), the scale bytes 2^(ec - 11)
};

// Activation exponents of the fp16 family (SFD2_PREC_F16 / F16C): the stored tensor of group g is 2^act_exp[g] times the network's
// tensor.  ReLU commutes with a positive factor and the factor is a power of two, so folding 2^(e_out - e_in) into a layer's scale
// and 2^e_out into its shift changes nothing but where the tensor sits in the number formats: exact in fp32, and it is what keeps
// the compensated mode's fixed-point-like corr units (e4m3 of x / 4 and of the fp16 residual * 512, tensors saturating at 1792)
// in their sweet spot whatever scale the checkpoint's activations have.  Set by sfd2_calibrate_range (api_weights.hip).
enum { AE_CONV1A, AE_CONV1B, AE_CONV2A, AE_CONV2B, AE_CONV3A, AE_TRUNK /* conv3b's and every ResBlock's output: one skip path */,
       AE_T1_0, AE_T1_1, AE_T1_2, AE_T2_0, AE_T2_1, AE_T2_2, AE_PA0, AE_DA0, AE_COUNT };

struct ActInfo { const void *p; int f32; int planar; int c, pitch, h, w; const void *pc = nullptr; /* corr plane (f16c) */ int exp2 = 0; /* stored = value * 2^exp2 (activation exponents of the fp16 family) */ bool absent = false; /* stays on chip on the path taken */ bool fmt6 = false; /* pc holds fp6 half-records (option "fp6_acts") */ bool r1 = false; /* pc holds one residual byte per channel, c bytes per pixel (option "trunk_r1") */ };

struct sfd2_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_jobs = nullptr;      // guards reuse of the pinned job descriptors below
    // host images go through a copy stream into one of two staging slots, so the upload of image i + 1 overlaps the
    // network of image i when the caller runs extracts back to back (SFD2_FLAG_ASYNC + pinned host memory)
    hipStream_t copy_stream = nullptr;
    // (SFD2_IMG_SLOTS slots, not two: an upload that has to WAIT for its slot's previous reader stalls the copy engine's queue in front of every other
    //  context's uploads -- two pipelined contexts on one device then run at half the rate of one (CHANGELOG round 5) -- so the ring is longer than the images
    //  a driver keeps in flight per context)
#define SFD2_IMG_SLOTS 4
    hipEvent_t ev_copied[SFD2_IMG_SLOTS] = {}, ev_img_free[SFD2_IMG_SLOTS] = {};
    DevBuf img2[SFD2_IMG_SLOTS];
    int img_slot = 0, img_slot_used = -1;
    void *pin_jobs = nullptr;
    size_t pin_cap = 0;
    bool weights_loaded = false;
    bool counters_clean = false;       // the kernel in front of the selection cleared the counters (pb_heads_heat_kernel)
    bool has_sta = false;              // ConvSta present in the loaded state_dict (absent for require_stability=False models)
    int opt_alias = 1;                 // sfd2_set_option "alias"
    int act_exp[AE_COUNT] = {};        // activation exponents of the fp16 family (above)
    float act_max[AE_COUNT] = {};      // what the calibration measured (largest |x| of the group's tensors on the calibration image)
    int opt_range_fallback = 1;        // sfd2_set_option "range_fallback": a synchronous f16c extract that saturated a tensor is re-run in f16x3
    int range_fallbacks = 0;           // how often that happened
    int in_fallback = 0;               // set around the re-run
    int prof_step_entry = 0;           // profile step at the entry of the running extraction: the repeat starts there again
    float range_hist[SFD2_RS_COUNT] = {};   // maxima (stored units) folded away from the device words by a fallback's reset
    int opt_auto_range = 1;            // sfd2_set_option "auto_range": sfd2_load_weights calibrates the exponents on a built-in probe image
    std::vector<float> h_sta_w;        // ConvSta filters as loaded (the uploaded copy carries 2^-act_exp[AE_TRUNK])
    int opt_cu_limit = 0;              // sfd2_set_option "cu_limit": persistent kernels of THIS context launch at most so many blocks
    int fuse_det = 0;                  // sfd2_set_option "fuse_det"
    int use_graphs = 0;                // sfd2_set_option "graphs"
    int alias_now = 0;                 // set per call
    int x3_fast_rb_now = 0;            // set per call: f16x3 ResBlocks on the streaming three-pass 1x1 kernel (not on the parity entry point:
                                       // the grouped conv's output then exists as planes only)
    DevBuf x3_chain2;                  // second buffer of the plane chain (a layer never writes the planes it reads)
    DevBuf x3_rb_planes[3];            // a ResBlock's input, conv1's and the grouped conv's outputs as hi / lo' planes
    const void *x3_pre_src = nullptr;  // set by a producer that wrote its output as planes too: the fp32 tensor they belong to ...
    const half_t *x3_pre_hi = nullptr, *x3_pre_lo = nullptr;   // ... and the planes (consumed by the next convf on that tensor)
    int x3_s2d_out_now = 0;            // set around conv2a's convf call: its planes are stored space-to-depth for conv2b_s2d_kernel<x3> (option "s2d")
    int x3_planes_out_now = 0;         // set around a convf call: the 3x3 layer writes hi / lo' planes INTO x3_chain instead of fp32
    DevBuf x3_chain;                   // planes handed from conv3a to conv3b (throughput path of f16x3)
    int opt_fuse_post = 1;             // sfd2_set_option "fuse_post": heads -> heat map -> NMS in one kernel on the extract path
    int skip_head_now = 0;             // set per call: run_network leaves the detector soft-max to the fused NMS kernel
    int opt_sparse_desc = 1;           // sfd2_set_option "sparse_desc": extract path runs convDb on the sampled corner pixels only
    int skip_db_now = 0;               // set per call: run_network leaves convDb to the sparse descriptor head
    int skip_da3_now = 0;              // set per call: run_network leaves convDa.3 to the sparse descriptor path (sparse_da3_kernel)
    const half_t *da0_cur = nullptr;   // convDa.0 output of the last fp16 network pass
    DevBuf da3_sparse;                 // [sel_cap][4][256] fp16: convDa.3 on the sampled corner pixels
    int opt_fp6_acts = 1;              // sfd2_set_option "fp6_acts": the corr records of the three tensors only conv3x3_pp<comp> reads (conv1b's, conv2b's,
                                       // conv3a's output) as block-scaled fp6 half-records, their consumers' corr MFMAs fp6 x fp6 (33.5 cycles instead of 66)
    int net_error = 0;                 // set by a layer helper of run_network that cannot return an error itself; run_network returns -1 and clears it
    int opt_trunk_r1 = 1;              // sfd2_set_option "trunk_r1": conv3b's output and the outputs of ResBlocks 0 and 1 carry ONE correction byte per channel (the
                                       // residual) instead of the (residual, value) unit: 3 bytes per channel through the HBM-bound ResBlock kernels instead of 4
    int opt_s2d = 1;                   // sfd2_set_option "s2d": on the throughput path conv2a stores its output space-to-depth and conv2b runs as a stride-1 layer
                                       // over it (conv2b_s2d_kernel.hip) instead of conv3x3_rf<2,comp>
    int opt_fp6_filters = 0;           // sfd2_set_option "fp6_filters": conv3x3_pp<comp> takes its corr filters as block-scaled fp6 (fp8 x fp6 MFMA).
                                       // Measured: conv3b 212.0 -> 212.1 us, extract 1.5188 -> 1.5165 ms, descriptors <=5.2e-4 (<=4.9e-4 without): the
                                       // mixed-format MFMA's shorter issue time in the probe does not show in the layer; off by default, kept as the
                                       // packing / layout groundwork for fp6 on both sides
    int opt_x3_pp = 1;                 // sfd2_set_option "x3_pp": SFD2_PREC_F16X3 runs its 3x3 stride-1 layers on conv3x3_pp (pre-split planes, three passes)
    int opt_auto_margin = 1;           // sfd2_set_option "auto_margin": sfd2_load_weights measures SFD2_PREC_F16C's descriptor error on the built-in probe against the
                                       // SFD2_PREC_F32 pass it runs anyway and, above SFD2_MARGIN_TARGET, turns on the accuracy options that bring it back (api_weights.hip)
    float margin_err[4] = {-1.0f, -1.0f, -1.0f, -1.0f};   // probe error with the options as set / rb_inner = 0 / comp_heads = 1 / both (-1: not measured)
    int margin_choice = -1;            // which of the four the context now runs (-1: no self-check has run)
    int opt_c3b_plain = 0;             // effective: conv3b (SFD2_PREC_F16C) runs its K loop over the hi plane of conv3a's output only -- no correction chunks (the
                                       // output's corr bytes still come from the fp32 accumulators) -- and conv3a then writes no corr plane.  -62 us of 184 at 1600x1200
                                       // for ~ x 1.4 on the descriptor error (tools/relax_study.py), so it is never on unverified:
    int user_c3b_plain = -1;           // sfd2_set_option "c3b_plain": -1 (default) = on only when the load-time self-check measured the probe inside SFD2_MARGIN_TARGET
                                       // WITH it (option "auto_margin"; off otherwise), 1 = on, 0 = off
    bool margin_done = false, margin_pending = false;   // the self-check ran on the loaded weights / waits for the context to enter SFD2_PREC_F16C
    float relax_err = -1.0f;           // the probe's error with "c3b_plain" on and the other options as set (-1: not measured)
    bool user_set_rb_inner = false, user_set_comp_heads = false;   // the caller set the key explicitly: the self-check reports, but does not override it
    int user_rb_inner = 2, user_comp_heads = 0;   // what sfd2_set_option last asked for: the self-check of a LATER sfd2_load_weights starts from these, not from its own earlier choice
    int opt_x3_desc16 = 0;             // sfd2_set_option "x3_desc16": SFD2_PREC_F16X3 on sfd2_extract with the DESCRIPTOR branch (convDa.0, convDa.3 at the sampled corners,
                                       // convDb) in plain fp16 on the backbone output's hi plane: the key points are this mode's own, the descriptors carry the
                                       // fp16 head's error only (<= 1e-3: north_star's tolerance, not this mode's 2e-5)
    int x3_desc16_now = 0;             // set per call by run_network: convDa.0's output is the fp16 tensor in x3_da0_planes (da0_cur)
    DevBuf x3_planes;                  // the input of such a layer as hi / lo' planes
    DevBuf x3_da0_planes;              // convDa.0's output as planes (sparse descriptor head of f16x3)
    DevBuf db_sparse;                  // [sel_cap][4][128] fp32: convDb on the sampled corners (f16x3)
    int opt_sparse_da3 = 1;            // sfd2_set_option "sparse_da3": with the sparse descriptor head, convDa.3 on the sampled corners only
    int skip_pb_now = 0;               // set per call: run_network leaves convPb to the fused detector head
    int opt_fuse_pb = 1;               // sfd2_set_option "fuse_pb": convPb inside the fused detector-head / heat-map kernel
    const half_t *pa_cur = nullptr;    // convPa.3 output of the last fp16 network pass
    const half_t *da_cur = nullptr;    // convDa.3 output of the last fp16 network pass
    int opt_comp_rb = 1;               // sfd2_set_option "comp_rb": SFD2_PREC_F16C compensates the ResBlocks too (0: fused fp16 ResBlock kernel)
    int opt_rb_inner = 2;              // sfd2_set_option "rb_inner": SFD2_PREC_F16C ResBlocks, 1 = t2 (the grouped conv's output) stored as plain
                                       // fp16, 2 (default) = t1 and t2, 0 = both compensated; the filters stay compensated either way
                                       // (second fp16 pass with their residuals).  Measured at 1600x1200: 1.82 / 1.75 / 1.67 ms per
                                       // extract for 0 / 1 / 2, descriptors <= 3.5e-4 / 3.8e-4 / 4.9e-4 over the BASELINE geometries.
    int opt_fuse_rb23 = 1;             // sfd2_set_option "fuse_rb23": with rb_inner = 2, ResBlock.conv2 + conv3 + residual in one kernel (t2 stays in LDS)
    int opt_comp_det = 0;              // sfd2_set_option "comp_det": SFD2_PREC_F16C compensates the detector branch's 3x3 layers (convPa.0, convPa.3) only
    int opt_comp_heads = 0;            // sfd2_set_option "comp_heads": SFD2_PREC_F16C compensates the 3x3 layers of the two head branches too
    int opt_no_rf_c = 0;               // sfd2_set_option "no_rf_c": conv2b on conv_igemm2<comp> instead of conv3x3_rf<comp> (A/B switch)
    int opt_generic_c = 0;             // sfd2_set_option "generic_c": SFD2_PREC_F16C layers on the generic reference kernel (tests)
    int opt_branches = 0;              // sfd2_set_option "branches": detector branch on a second stream beside the descriptor branch
    int opt_sta_side = 0;              // sfd2_set_option "sta_side": ConvSta on the side stream beside the head branches' 3x3 layers (throughput path)
    hipStream_t side_stream = nullptr; // the detector branch (convPa.0 -> convPa.3 -> convPb -> detector_head)
    hipStream_t cur_stream = nullptr;  // stream the conv()/ProfScope helpers launch on (main or side)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    struct GraphEntry *graphs = nullptr;   // hipGraph cache of sfd2_extract_match (see below)
    int n_graphs = 0;
    unsigned long long graph_clock = 0;
    // weights
    ConvW c1a, c1b, c2a, c2b, c3a, c3b, rb1[3], rb2[3], rb3[3], pa0, pa3, da0, da3, pb, db;
    DevBuf sta_w16;                                // ConvSta filters times 2^-act_exp[AE_TRUNK]: what the fp16 family launches with
    DevPtr range_stat;                             // (a view: the words live behind the zero page) SFD2_RS_COUNT x SFD2_RANGE_SUB words: running maxima of the compensated mode's stored tensors (sticky until read with reset)
    DevPtr range_hist_dev;                         // (a view behind range_stat) SFD2_RS_COUNT words: maxima folded away from the running words on the device
    DevPtr extract_rec;                            // (a view behind range_hist_dev) the four words of sfd2_extract_record_async
    DevBuf range_scratch;                          // AE_COUNT words: absolute maxima of a calibration pass
    DevBuf sta_w, sta_b, zero_page, w1b_fused;   // w1b_fused: conv1b filters as [9][64][64] for the fused stem
    DevBuf w1b_stem_c6;                            // ... with the corr fragments as fp6 strings + scale byte (option "fp6_acts")
    DevBuf w1b_stem_c;                             // the same as register fragments (hi K slices + corr) for the compensated fused stem
    DevBuf w1b_stem_x3;                            // ... with the lo' fragments (fp16 of (w - fp16(w)) * 2^11) in place of the corr fragment: f16x3
    int fuse = 1;                                  // fused kernels on the extract path (SFD2_NO_FUSE=1 disables)
    int fuse_now = 0;                              // set per call: sfd2_det keeps every intermediate readable
    // strict fp32 mode
    int precision = SFD2_PREC_F16;
    ConvW f1a, f1b, f2a, f2b, f3a, f3b, frb1[3], frb2[3], frb3[3], fpa0, fpa3, fda0, fda3, fpb, fdb;
    DevBuf g1a, g1b, g2a, g2b, g3a, g3b, grt1[3], grt2[3], gro[3], gpa0_o, gpa_o, gda0_o, gda_o;   // fp32 NHWC activations
    // geometry of the current workspace
    int H = 0, W = 0, H2 = 0, W2 = 0, H4 = 0, W4 = 0, H8 = 0, W8 = 0;
    // activations (NHWC fp16 unless noted)
    DevBuf img, a1a, a1b, a2a, a2b, a3a, a3b, rt1[3], rt2[3], ro[3], pa0_o, pa_o, da0_o, da_o;
    DevBuf logits /*f32 [P8][128]*/, draw /*f32 [P4][128]*/, sta /*f32 [3][P4]*/, score /*f32*/, heat /*f32*/;
    DevBuf stab /*f32 [H][W]*/, desc_nchw, tmp_f32;
    // selection
    DevBuf cand, bnd, sel, sorted, counters, kpts, kscores, kdesc;
    DevBuf kdesc64;                    // SFD2_FLAG_DESC_STORE64: the descriptors as float64 [128][capacity] on their way to a host buffer
    DevBuf g_keys, g_state0, g_state1, g_kept;   // greedy NMS (extract.py variant)
    int cand_cap = 0;
    int last_sel_cap = 0;
    float *kpts_cur = nullptr, *kscores_cur = nullptr;   // where the last selection wrote its key points
    // scale pyramid staging (sfd2_extract_multiscale)
    DevBuf arena;   // aliased activation slots of the throughput path (run_network)
    DevBuf img_u8_packed;              // SFD2_FLAG_IMG_U8_X: the image as three bytes per pixel (unpack_rgbx_kernel)
    DevBuf img_scaled, ms_kp, ms_sc, ms_de, ms_keys, ms_sorted, ms_cnt;
    unsigned int ms_cand_seen[8] = {};
    int ms_cand_cap[8] = {};
    // matcher
    DevBuf m_stage, m_hi0, m_lo0, m_hi1, m_lo1, m_part_f, m_part_i, m_red, m_jobs, m_fins, m_out_m, m_out_s, m_rkeys;
    sfd2_timings tim = {};
    std::map<std::string, ActInfo> acts;
    // per-launch profiling (sfd2_set_profiling)
    int prof_max_steps = 0, prof_step = 0, prof_slot = 0;
    std::vector<hipEvent_t> prof_ev;          // [max_steps][PROF_SLOTS][2]
    std::vector<sfd2_layer_timing> prof_tab;  // slot -> descriptor + accumulators
    std::vector<int> prof_used;               // [max_steps] slots recorded in that step
    std::vector<int> prof_row;                // [max_steps][PROF_SLOTS] -> row of prof_tab
    std::string prof_filter;                  // only kernel labels containing this are timed
};

void graphs_release(sfd2_ctx *c);
#define PROF_SLOTS 48
struct ProfScope {   // records an event pair around one launch when profiling is on
    sfd2_ctx *c; int slot;
    ProfScope(sfd2_ctx *c_, const char *name, const char *kernel, double flops, double bytes) : c(c_), slot(-1)
    {
        if (c->prof_max_steps <= 0 || c->prof_step >= c->prof_max_steps || c->prof_slot >= PROF_SLOTS) return;
        if (!c->prof_filter.empty() && strstr(kernel, c->prof_filter.c_str()) == nullptr) return;
        slot = c->prof_slot++;
        int row = -1;  // table rows are keyed by stage name (extract and match steps interleave)
        for (size_t i = 0; i < c->prof_tab.size(); ++i)
            if (strncmp(c->prof_tab[i].name, name, sizeof(c->prof_tab[i].name) - 1) == 0) { row = (int)i; break; }
        if (row < 0) {
            c->prof_tab.push_back(sfd2_layer_timing{});
            row = (int)c->prof_tab.size() - 1;
            snprintf(c->prof_tab[row].name, sizeof(c->prof_tab[row].name), "%s", name);
            snprintf(c->prof_tab[row].kernel, sizeof(c->prof_tab[row].kernel), "%s", kernel);
        }
        c->prof_tab[row].flops = flops;
        c->prof_tab[row].bytes = bytes;
        c->prof_row[(size_t)c->prof_step * PROF_SLOTS + slot] = row;
        (void)hipEventRecord(c->prof_ev[((size_t)c->prof_step * PROF_SLOTS + slot) * 2], c->cur_stream);
    }
    void cancel()   // the launch did not happen (a launcher declined the geometry): give the slot back
    {
        if (slot >= 0 && slot == c->prof_slot - 1) c->prof_slot--;
        slot = -1;
    }
    ~ProfScope()
    {
        if (slot >= 0) (void)hipEventRecord(c->prof_ev[((size_t)c->prof_step * PROF_SLOTS + slot) * 2 + 1], c->cur_stream);
    }
};
static inline void prof_step_begin(sfd2_ctx *c) { c->prof_slot = 0; }
static inline void prof_step_end(sfd2_ctx *c)
{
    if (c->prof_max_steps > 0 && c->prof_step < c->prof_max_steps) {
        c->prof_used[c->prof_step] = c->prof_slot;
        c->prof_step++;
    }
}

// api_weights.hip
int apply_act_exponents(sfd2_ctx *c);                   // re-uploads the fp16 family's scale / shift arrays with c->act_exp folded in
// api_core.hip
int read_range_status(sfd2_ctx *c, sfd2_range_status *out, int reset);
// after a synchronous extraction in SFD2_PREC_F16C: 1 = a compensated tensor saturated and the call is to be repeated in SFD2_PREC_F16X3
// (the device words are folded into the context's history and cleared), 0 = fine / not applicable, -1 = error
int range_wants_fallback(sfd2_ctx *c);
int range_fold_before_sync_extract(sfd2_ctx *c);        // earlier asynchronous calls' maxima -> device-side history, running words cleared (no host sync)
int sfd2_margin_selfcheck_if_pending(sfd2_ctx *c);    // api_weights.hip: the F16C self-check of a context loaded in another precision
int reset_range_records(sfd2_ctx *c);                   // the exponents changed: running words, device and host history cleared
struct FallbackScope {      // the repeat: strict arithmetic, no recursion
    sfd2_ctx *c; int prec;
    explicit FallbackScope(sfd2_ctx *c_) : c(c_), prec(c_->precision)
    {
        c->precision = SFD2_PREC_F16X3; c->in_fallback = 1; c->range_fallbacks++;
        if (c->prof_step > c->prof_step_entry) c->prof_step = c->prof_step_entry;   // the saturated pass is not a profiled step: its slots are recorded again
    }
    ~FallbackScope() { c->precision = prec; c->in_fallback = 0; }
};
// api_network.hip
void set_path(sfd2_ctx *c, bool parity_entry);          // which kernels / buffers the next network pass uses; call before ensure_workspace
int ensure_workspace(sfd2_ctx *c, int H, int W);
int run_network(sfd2_ctx *c, const float *img_dev, int normalise);
int convf(sfd2_ctx *c, const char *name, const ConvW &L, const DevPtr &in, int H, int W, const DevPtr &out, int Ho, int Wo, int relu,
          const float *res = nullptr);
// api_extract.hip
int copy_out(sfd2_ctx *c, void *dst, const void *src_dev, size_t bytes, int dst_on_device);
