// libsfd2hip: BatchNorm folding and filter packing (sfd2_load_weights).
#include "sfd2_ctx.h"

// ------------------------------------------------------------------------------------------ weights
struct TView {
    const float *d;
    std::vector<int64_t> shape;
    std::vector<int> ec;       // conv filters after normalise_filters: d holds w[oc] * 2^-ec[oc]; fold_scale_shift puts 2^ec[oc] into the scale
    size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};
typedef std::map<std::string, TView> TMap;

static const TView *find_t(const TMap &m, const std::string &k)
{
    auto it = m.find(k);
    return it == m.end() ? nullptr : &it->second;
}

static int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st)
{
    HIPCHECK(b.ensure(bytes));
    HIPCHECK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

static float ec_factor(const TMap &m, const std::string &conv, int c)
{
    const TView *w = find_t(m, conv + ".weight");
    return (w && (size_t)c < w->ec.size()) ? std::ldexp(1.0f, w->ec[c]) : 1.0f;
}

// the folded epilogue constants of a layer: host copies stay with the layer (the activation exponents of the fp16 family are
// applied to them later: apply_act_exponents)
static int set_scale_shift(sfd2_ctx *c, ConvW &L, const std::vector<float> &sc, const std::vector<float> &sh)
{
    L.h_scale = sc;
    L.h_shift = sh;
    if (upload(L.scale, sc.data(), sc.size() * sizeof(float), c->stream)) return -1;
    return upload(L.shift, sh.data(), sh.size() * sizeof(float), c->stream);
}

// y = scale * conv_nobias(x) + shift  with conv bias and BatchNorm(eval, eps 1e-5) folded:
//   BN(affine=False): (x + b - mean) / sqrt(var + eps)                         nets/sfd2.py:58-65
//   BN(affine):       gamma * (x + b - mean) / sqrt(var + eps) + beta           nets/sfd2.py:286-296, :25-55
static int fold_scale_shift(const TMap &m, const std::string &conv, const std::string &bn, int cout, int cout_pad,
                            std::vector<float> &scale, std::vector<float> &shift)
{
    scale.assign(cout_pad, 1.0f);
    shift.assign(cout_pad, 0.0f);
    const TView *bias = find_t(m, conv + ".bias");
    if (bias && (int)bias->numel() != cout) return fail("bad bias shape for " + conv);
    if (bn.empty()) {
        for (int c = 0; c < cout; ++c) { scale[c] = ec_factor(m, conv, c); shift[c] = bias ? bias->d[c] : 0.0f; }
        return 0;
    }
    const TView *mean = find_t(m, bn + ".running_mean"), *var = find_t(m, bn + ".running_var");
    const TView *gamma = find_t(m, bn + ".weight"), *beta = find_t(m, bn + ".bias");
    if (!mean || !var) return fail("missing BatchNorm statistics: " + bn);
    if ((int)mean->numel() != cout || (int)var->numel() != cout) return fail("bad BatchNorm shape: " + bn);
    for (int c = 0; c < cout; ++c) {
        const float inv = 1.0f / std::sqrt(var->d[c] + 1e-5f);
        const float a = gamma ? gamma->d[c] * inv : inv;
        const float b = bias ? bias->d[c] : 0.0f;
        scale[c] = a * ec_factor(m, conv, c);      // (a power of two: exact)
        shift[c] = (beta ? beta->d[c] : 0.0f) + (b - mean->d[c]) * a;
    }
    return 0;
}

// Per-output-channel power-of-two normalisation of a conv layer's filters: w'[oc] = w[oc] * 2^-ec[oc] with max|w'[oc]| in [1, 2),
// and 2^ec[oc] goes into the folded scale.  Exact in every mode (products and fp32 sums scale by the same power of two, the
// epilogue's acc * scale is the same number), so a well-conditioned checkpoint gives the same bits as without it.  What it buys:
// a TRAINED checkpoint has no reason to keep its filters in fp16's comfortable range -- weight decay under a BatchNorm shrinks a
// channel's filter freely (running_var follows), and below 6e-5 the fp16 parts go subnormal, the e4m3 correction units of a
// channel 2^-9 of the layer's largest vanish altogether.  After the normalisation every channel uses the full range of both.
static void normalise_filters(TMap &m, const std::string &conv, std::vector<std::vector<float>> &owned)
{
    auto it = m.find(conv + ".weight");
    if (it == m.end() || it->second.shape.size() != 4) return;
    TView &w = it->second;
    const size_t cout = (size_t)w.shape[0], per = w.numel() / std::max<size_t>(cout, 1);
    owned.emplace_back(w.d, w.d + w.numel());
    std::vector<float> &o = owned.back();
    w.ec.assign(cout, 0);
    for (size_t oc = 0; oc < cout; ++oc) {
        float mx = 0.0f;
        for (size_t i = 0; i < per; ++i) mx = std::max(mx, std::fabs(o[oc * per + i]));
        if (!(mx > 0.0f) || !std::isfinite(mx)) continue;
        int e = 0;
        (void)std::frexp(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)  ->  mx * 2^-(e - 1) in [1, 2)
        e = std::max(-100, std::min(100, e - 1));
        w.ec[oc] = e;
        for (size_t i = 0; i < per; ++i) o[oc * per + i] = std::ldexp(o[oc * per + i], -e);
    }
    w.d = o.data();
}

// OCP fp8 e4m3fn, round to nearest even, saturating at +-448 (the filters' corr units of SFD2_PREC_F16C)
static unsigned char f32_to_e4m3(float f)
{
    const unsigned char sign = std::signbit(f) ? 0x80 : 0x00;
    float a = std::fabs(f);
    if (!(a == a)) return sign | 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < 0x1p-10f) return sign;                     // below half of the smallest subnormal (2^-9); the tie rounds to even = 0
    int e;
    (void)std::frexp(a, &e);                           // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;                                    // a in [2^ex, 2^(ex+1))
    if (ex < -6) ex = -6;                              // subnormal range shares the exponent of the smallest normal
    const float q = std::ldexp(a, 3 - ex);             // units of 2^(ex-3): 8..16 for normals, 0..8 for subnormals
    float r = std::nearbyint(q);                       // default rounding mode: to nearest even
    int mant = (int)r, be = ex + 7;
    if (ex == -6 && mant < 8) return sign | (unsigned char)mant;      // subnormal (biased exponent 0)
    if (mant == 16) { mant = 8; ++be; }
    if (be > 15 || (be == 15 && mant - 8 > 6)) return sign | 0x7e;
    return sign | (unsigned char)((be << 3) | (mant - 8));
}

// e2m3 (fp6: sign, 2 exponent bits with bias 1, 3 mantissa bits; subnormal step 0.125, largest value 7.5), round to nearest even,
// saturating.  The operand format of v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz / blgp = 2 (codes checked on the part:
// tools/probe/mfma_fp6_layout.hip).
static unsigned char f32_to_e2m3(float v)
{
    const unsigned char sign = std::signbit(v) ? 0x20 : 0;
    const float a = std::fabs(v);
    if (!(a == a) || a >= 7.75f) return sign | 31;
    if (a < 1.0f) {
        const int m = (int)std::nearbyint(a * 8.0f);           // 0 .. 8 (8 = the smallest normal)
        return sign | (unsigned char)m;
    }
    int e;
    (void)std::frexp(a, &e);
    int ex = e - 1;                                            // a in [2^ex, 2^(ex+1)), ex = 0 .. 2
    int mant = (int)std::nearbyint(std::ldexp(a, 3 - ex)) - 8; // 0 .. 8
    if (mant == 8) { mant = 0; ++ex; }
    if (ex > 2) return sign | 31;
    return sign | (unsigned char)(((ex + 1) << 3) | mant);
}

// scale exponent b0 of a layer's corr filters: the largest power of two with max|w| * 2^b0 <= 448
static int corr_b0(const float *w, size_t n)
{
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, std::fabs(w[i]));
    if (!(mx > 0.0f) || !std::isfinite(mx)) return 0;
    int b0 = (int)std::floor(std::log2(448.0f / mx));
    while (std::ldexp(mx, b0) > 448.0f) --b0;
    return std::max(-40, std::min(40, b0));
}

static int pack_igemm(sfd2_ctx *c, const TMap &m, ConvW &L, const std::string &conv, const std::string &bn, int cin,
                      int cout, int ks, int stride)
{
    const TView *w = find_t(m, conv + ".weight");
    if (!w) return fail("missing tensor: " + conv + ".weight");
    if (w->shape.size() != 4 || w->shape[0] != cout || w->shape[1] != cin || w->shape[2] != ks || w->shape[3] != ks)
        return fail("bad shape for " + conv + ".weight");
    const int cout_pad = (cout + 63) / 64 * 64;
    L.cin = cin; L.cout = cout; L.cout_pad = cout_pad; L.ks = ks; L.stride = stride;
    const int cc = conv_igemm_chunk(ks, stride, cout_pad, cin);   // 32 or 64 input channels per packed tile
    const int T = ks * ks, nch = cin / cc;
    std::vector<half_t> pk((size_t)nch * T * cout_pad * cc, (half_t)0.0f);
    for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < T; ++t)
            for (int oc = 0; oc < cout; ++oc)
                for (int k = 0; k < cc; ++k) {
                    const float v = w->d[(((size_t)oc * cin + ch * cc + k) * ks + t / ks) * ks + t % ks];
                    pk[(((size_t)ch * T + t) * cout_pad + oc) * cc + k] = (half_t)v;
                }
    std::vector<float> sc, sh;
    if (fold_scale_shift(m, conv, bn, cout, cout_pad, sc, sh)) return -1;
    if (upload(L.w, pk.data(), pk.size() * sizeof(half_t), c->stream)) return -1;
    if (set_scale_shift(c, L, sc, sh)) return -1;
    if (ks == 1 && stride == 1 && cin == 256 && cout == 256) {
        std::vector<half_t> rm((size_t)256 * 256);
        for (size_t i = 0; i < rm.size(); ++i) rm[i] = (half_t)w->d[i];
        if (upload(L.wrm, rm.data(), rm.size() * sizeof(half_t), c->stream)) return -1;
    }
    {   // SFD2_PREC_F16C: 32-wide chunks of the fp16 filters, then the corr units in the same geometry
        const int nch32 = cin / 32;
        const size_t plane = (size_t)nch32 * T * cout_pad * 32;
        std::vector<unsigned short> pc(2 * plane, 0);
        const int b0 = corr_b0(w->d, w->numel());
        L.sbyte = 127 - SFD2_C_XL_SHIFT - b0;
        for (int ch = 0; ch < nch32; ++ch)
            for (int t = 0; t < T; ++t)
                for (int oc = 0; oc < cout; ++oc)
                    for (int k = 0; k < 32; ++k) {
                        const float v = w->d[(((size_t)oc * cin + ch * 32 + k) * ks + t / ks) * ks + t % ks];
                        const half_t h = (half_t)v;
                        const size_t o = (((size_t)ch * T + t) * cout_pad + oc) * 32 + k;
                        unsigned short hb;
                        std::memcpy(&hb, &h, 2);
                        pc[o] = hb;
                        const unsigned char w8 = f32_to_e4m3(std::ldexp(v, b0));
                        const unsigned char l8 = f32_to_e4m3(std::ldexp(v - (float)h, b0 + 11));
                        pc[plane + o] = (unsigned short)(w8 | (l8 << 8));   // pairs with the pixel unit (residual byte, value byte)
                    }
        if (upload(L.wc, pc.data(), pc.size() * 2, c->stream)) return -1;
        if (ks == 3 && stride == 1 && cout_pad % 128 == 0 && cin % 64 == 0) {
            // conv3x3_pp<comp>: the corr filter rows as fp6.  A row (64 B = the filters of one pixel record's 64 unit bytes j: j even ->
            // w of channel j / 2, pairing with the residual byte; j odd -> (w - fp16(w)) * 2^11, pairing with the value byte) becomes
            // two 24-byte strings of 32 six-bit codes, string h = bytes j = 32 h .. 32 h + 31, stored where lane half h of the kernel's
            // fragment read finds them: its first 16 bytes in the row's 16-byte slot h, the other 8 in slot 2 + h (the rest is padding).
            // One power-of-two scale per OUTPUT CHANNEL (all taps, all input channels): 2^ec >= max|w| / 7.5; its E8M0 byte goes to
            // the MFMA's A-side scale operand lane by lane.  fp8 x fp6 issues in 32 ns where fp8 x fp8 takes 37-41
            // (profiles/r03k_mfma_f8f6f4_probe.txt); descriptors on the CPU twin 4.1e-4 against 4.0e-4 (profiles/r03k_error_budget_fp6.txt).
            //
            // pix6 (option "fp6_acts"): the pixel records are fp6 half-records too (sfd2_epi16_fp6): position p of string h = code 2 j + (p & 1),
            // j = p / 2 <-> channel 8 (j / 4) + 4 h + (j % 4) of the chunk; p even pairs with the pixel's residual code (lo' = (x - hi) * 2^11 over
            // the pixel block's scale), p odd with its value code, so the filter side carries w, then lo'_w = (w - fp16(w)) * 2^11, and 2^-11 moves
            // into the scale byte: 2^(ec - 11).
            std::vector<int> ec(cout_pad, 0);
            for (int pix6 = 0; pix6 < 2; ++pix6) {
            std::vector<unsigned short> p6(2 * plane, 0);
            std::memcpy(p6.data(), pc.data(), plane * 2);
            std::vector<int> sa(2 * (size_t)cout_pad, 0x7f7f7f7f);      // [shift | scale bytes]
            std::memcpy(sa.data(), sh.data(), (size_t)cout_pad * sizeof(float));
            for (int oc = 0; oc < cout; ++oc) {
                float mx = 0.0f;
                for (size_t i = 0; i < (size_t)cin * T; ++i) mx = std::max(mx, std::fabs(w->d[(size_t)oc * cin * T + i]));
                int e = (mx > 0.0f && std::isfinite(mx)) ? (int)std::ceil(std::log2(mx / 7.5f)) : 0;
                while (std::ldexp(mx, -e) > 7.5f) ++e;
                e = std::max(-40, std::min(40, e));
                ec[oc] = e;
                sa[cout_pad + oc] = ((127 - (pix6 ? 11 : SFD2_C_XL_SHIFT) + e) & 255) * 0x01010101;
            }
            for (int ch = 0; ch < nch32; ++ch)
                for (int t = 0; t < T; ++t)
                    for (int oc = 0; oc < cout; ++oc) {
                        unsigned char *row = reinterpret_cast<unsigned char *>(p6.data() + plane) + ((((size_t)ch * T + t) * cout_pad + oc) * 32) * 2;
                        for (int h = 0; h < 2; ++h) {
                            unsigned char str[24] = {0};
                            for (int p6i = 0; p6i < 32; ++p6i) {
                                const int j = 32 * h + p6i, jj = p6i >> 1;
                                const int k = pix6 ? 8 * (jj >> 2) + 4 * h + (jj & 3) : j >> 1;
                                const float v = w->d[(((size_t)oc * cin + ch * 32 + k) * ks + t / ks) * ks + t % ks];
                                const float x = (j & 1) ? std::ldexp(v - (float)(half_t)v, 11 - ec[oc]) : std::ldexp(v, -ec[oc]);
                                const unsigned int code = f32_to_e2m3(x);
                                const int bit = 6 * p6i;
                                str[bit >> 3] |= (unsigned char)(code << (bit & 7));
                                if ((bit & 7) > 2) str[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
                            }
                            std::memcpy(row + 16 * h, str, 16);
                            std::memcpy(row + 32 + 16 * h, str + 16, 8);
                        }
                    }
            if (upload(pix6 ? L.wc66 : L.wc6, p6.data(), p6.size() * 2, c->stream)) return -1;
            (pix6 ? L.h_sa66 : L.h_sa6) = sa;
            if (upload(pix6 ? L.sa66 : L.sa6, sa.data(), sa.size() * sizeof(int), c->stream)) return -1;
            }
        }
        if (ks == 1 && stride == 1 && cin == 256 && cout == 256) {
            std::vector<half_t> fh((size_t)256 * 256), fl(fh.size());
            std::vector<unsigned short> fc(fh.size());
            std::vector<unsigned char> fr(fh.size() * 2);      // `wfr`: the corr units of a lane as [16 x w8][16 x lo_w8] (conv1x1_c256_c<.., 2, ..>)
            for (int wv = 0; wv < 8; ++wv)
                for (int cc = 0; cc < 8; ++cc)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 16; ++e) {
                            const int row = wv * 32 + (l & 31), col = cc * 32 + (e >> 3) * 16 + (l >> 5) * 8 + (e & 7);
                            const float v = w->d[(size_t)row * 256 + col];
                            const size_t o = (((size_t)wv * 8 + cc) * 64 + l) * 16 + e;
                            fh[o] = (half_t)v;
                            fl[o] = (half_t)((v - (float)(half_t)v) * 2048.0f);
                            fc[o] = (unsigned short)(f32_to_e4m3(std::ldexp(v, b0)) | (f32_to_e4m3(std::ldexp(v - (float)(half_t)v, b0 + 11)) << 8));
                            fr[(o - e) * 2 + e] = f32_to_e4m3(std::ldexp(v, b0));
                            fr[(o - e) * 2 + 16 + e] = f32_to_e4m3(std::ldexp(v - (float)(half_t)v, b0 + 11));
                        }
            if (upload(L.wfh, fh.data(), fh.size() * 2, c->stream)) return -1;
            if (upload(L.wfl, fl.data(), fl.size() * 2, c->stream)) return -1;
            if (upload(L.wfc, fc.data(), fc.size() * 2, c->stream)) return -1;
            if (upload(L.wfr, fr.data(), fr.size(), c->stream)) return -1;
            {   // `wf8l`: the filter residuals as e4m3((w - fp16(w)) * 2^(b0 + 11)) in rb23_c_kernel's K order for its scaled MFMAs over t2's value
                // bytes: [8 waves][4 j][64 lanes][32 B], byte n of a lane = channel 16 (4 j + n / 8) + 8 (lane / 32) + n % 8 of row wave * 32 + lane % 32
                std::vector<unsigned char> f8((size_t)8 * 4 * 64 * 32);
                for (int wv = 0; wv < 8; ++wv)
                    for (int j = 0; j < 4; ++j)
                        for (int l = 0; l < 64; ++l)
                            for (int n = 0; n < 32; ++n) {
                                const int row = wv * 32 + (l & 31), col = 16 * (4 * j + (n >> 3)) + 8 * (l >> 5) + (n & 7);
                                const float v = w->d[(size_t)row * 256 + col];
                                f8[(((size_t)wv * 4 + j) * 64 + l) * 32 + n] = f32_to_e4m3(std::ldexp(v - (float)(half_t)v, b0 + 11));
                            }
                if (upload(L.wf8l, f8.data(), f8.size(), c->stream)) return -1;
            }
        }
    }
    return 0;
}

static int pack_conv1a(sfd2_ctx *c, const TMap &m)
{
    const TView *w = find_t(m, "conv1a.0.weight");
    if (!w) return fail("missing tensor: conv1a.0.weight");
    if (w->shape.size() != 4 || w->shape[0] != 64 || w->shape[1] != 3 || w->shape[2] != 3 || w->shape[3] != 3)
        return fail("bad shape for conv1a.0.weight");
    ConvW &L = c->c1a;
    L.cin = 3; L.cout = 64; L.cout_pad = 64; L.ks = 3; L.stride = 1;
    // A fragment of mfma_32x32x16: lane l -> row (l & 31), k = (l >> 5) * 8 + j; k = kx * 4 + c within a filter row
    std::vector<half_t> pk((size_t)2 * 3 * 64 * 8, (half_t)0.0f);
    for (int ct = 0; ct < 2; ++ct)
        for (int ky = 0; ky < 3; ++ky)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int oc = ct * 32 + (lane & 31), g = lane >> 5;
                    const int kx = 2 * g + (j >> 2), ch = j & 3;
                    float v = 0.0f;
                    if (kx < 3 && ch < 3) v = w->d[(((size_t)oc * 3 + ch) * 3 + ky) * 3 + kx];
                    pk[(((size_t)ct * 3 + ky) * 64 + lane) * 8 + j] = (half_t)v;
                }
    std::vector<float> sc, sh;
    if (fold_scale_shift(m, "conv1a.0", "conv1a.1", 64, 64, sc, sh)) return -1;
    if (upload(L.w, pk.data(), pk.size() * sizeof(half_t), c->stream)) return -1;
    if (set_scale_shift(c, L, sc, sh)) return -1;
    {   // SFD2_PREC_F16C: the same fragments (hi) followed by fp16(w - hi) (lo)
        std::vector<half_t> pc(2 * pk.size(), (half_t)0.0f);
        for (int ct = 0; ct < 2; ++ct)
            for (int ky = 0; ky < 3; ++ky)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int oc = ct * 32 + (lane & 31), g = lane >> 5;
                        const int kx = 2 * g + (j >> 2), ch = j & 3;
                        float v = 0.0f;
                        if (kx < 3 && ch < 3) v = w->d[(((size_t)oc * 3 + ch) * 3 + ky) * 3 + kx];
                        const size_t o = (((size_t)ct * 3 + ky) * 64 + lane) * 8 + j;
                        pc[o] = (half_t)v;
                        pc[pk.size() + o] = (half_t)(v - (float)pc[o]);
                    }
        if (upload(L.wc, pc.data(), pc.size() * sizeof(half_t), c->stream)) return -1;
    }
    return 0;
}

static int pack_gconv(sfd2_ctx *c, const TMap &m, ConvW &L, const std::string &conv, const std::string &bn)
{
    const TView *w = find_t(m, conv + ".weight");
    if (!w) return fail("missing tensor: " + conv + ".weight");
    if (w->shape.size() != 4 || w->shape[0] != 256 || w->shape[1] != 8 || w->shape[2] != 3 || w->shape[3] != 3)
        return fail("bad shape for " + conv + ".weight (expected [256,8,3,3], groups=32)");
    L.cin = 256; L.cout = 256; L.cout_pad = 256; L.ks = 3; L.stride = 1;
    // A fragment of mfma_16x16x32: lane l -> row i = l & 15 (output channel of the pair), k = (l >> 4) * 8 + j;
    // k step s covers taps 2s, 2s+1: k = (tap - 2s) * 16 + (input channel of the pair)
    std::vector<half_t> pk((size_t)16 * 5 * 64 * 8, (half_t)0.0f), pl(pk.size(), (half_t)0.0f);
    for (int pair = 0; pair < 16; ++pair)
        for (int s = 0; s < 5; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int i = lane & 15, g = lane >> 4;
                    const int tap = 2 * s + (g >> 1);
                    const int oc = pair * 16 + i;
                    float v = 0.0f;
                    if (tap <= 8 && (i >> 3) == (g & 1)) v = w->d[(((size_t)oc * 8 + j) * 3 + tap / 3) * 3 + tap % 3];
                    pk[(((size_t)pair * 5 + s) * 64 + lane) * 8 + j] = (half_t)v;
                    pl[(((size_t)pair * 5 + s) * 64 + lane) * 8 + j] = (half_t)((v - (float)(half_t)v) * 2048.0f);
                }
    if (upload(L.wlk, pl.data(), pl.size() * sizeof(half_t), c->stream)) return -1;
    std::vector<float> sc, sh;
    if (fold_scale_shift(m, conv, bn, 256, 256, sc, sh)) return -1;
    if (upload(L.w, pk.data(), pk.size() * sizeof(half_t), c->stream)) return -1;
    if (set_scale_shift(c, L, sc, sh)) return -1;
    std::vector<half_t> cp((size_t)256 * 9 * 8);
    for (int oc = 0; oc < 256; ++oc)
        for (int tap = 0; tap < 9; ++tap)
            for (int j = 0; j < 8; ++j)
                cp[((size_t)oc * 9 + tap) * 8 + j] = (half_t)w->d[(((size_t)oc * 8 + j) * 3 + tap / 3) * 3 + tap % 3];
    if (upload(L.wgc, cp.data(), cp.size() * sizeof(half_t), c->stream)) return -1;
    {   // SFD2_PREC_F16C: corr fragments of v_mfma_scale_f32_16x16x128_f8f6f4 (gconv_c_kernel): [pair][step m][lane][32 B] --
        // lane -> out channel (lane & 15) of the pair, tap 4 * m + (lane >> 4); its 32 bytes = the 16 input channels of the
        // pair x (fp8 of w * 2^b0, fp8 of (w - fp16(w)) * 2^(b0 + 11)), zero outside the channel's own group
        const int b0 = corr_b0(w->d, w->numel());
        L.sbyte = 127 - SFD2_C_XL_SHIFT - b0;
        std::vector<unsigned short> pc((size_t)16 * 3 * 64 * 16, 0);
        for (int pair = 0; pair < 16; ++pair)
            for (int mm = 0; mm < 3; ++mm)
                for (int lane = 0; lane < 64; ++lane)
                    for (int ch = 0; ch < 16; ++ch) {
                        const int i = lane & 15, tap = 4 * mm + (lane >> 4);
                        const int oc = pair * 16 + i;
                        if (tap > 8 || (i >> 3) != (ch >> 3)) continue;
                        const float v = w->d[(((size_t)oc * 8 + (ch & 7)) * 3 + tap / 3) * 3 + tap % 3];
                        pc[(((size_t)pair * 3 + mm) * 64 + lane) * 16 + ch] =
                            (unsigned short)(f32_to_e4m3(std::ldexp(v, b0)) | (f32_to_e4m3(std::ldexp(v - (float)(half_t)v, b0 + 11)) << 8));
                    }
        if (upload(L.wc, pc.data(), pc.size() * 2, c->stream)) return -1;
    }
    return 0;
}

// ---- strict fp32 mode: the same folding, filters kept in fp32
static int pack_igemm_f32(sfd2_ctx *c, const TMap &m, ConvW &L, const std::string &conv, const std::string &bn, int cin,
                          int cout, int ks, int stride)
{
    const TView *w = find_t(m, conv + ".weight");
    if (!w) return fail("missing tensor: " + conv + ".weight");
    if (w->shape.size() != 4 || w->shape[0] != cout || w->shape[1] != cin || w->shape[2] != ks || w->shape[3] != ks)
        return fail("bad shape for " + conv + ".weight");
    const int cout_pad = (cout + 63) / 64 * 64;
    L.cin = cin; L.cout = cout; L.cout_pad = cout_pad; L.ks = ks; L.stride = stride;
    const int T = ks * ks, nch = cin / 32;
    std::vector<float> pk((size_t)nch * T * cout_pad * 32, 0.0f);
    for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < T; ++t)
            for (int oc = 0; oc < cout; ++oc)
                for (int k = 0; k < 32; ++k)
                    pk[(((size_t)ch * T + t) * cout_pad + oc) * 32 + k] =
                        w->d[(((size_t)oc * cin + ch * 32 + k) * ks + t / ks) * ks + t % ks];
    std::vector<float> sc, sh;
    if (fold_scale_shift(m, conv, bn, cout, cout_pad, sc, sh)) return -1;
    if (upload(L.w, pk.data(), pk.size() * sizeof(float), c->stream)) return -1;
    if (set_scale_shift(c, L, sc, sh)) return -1;
    return 0;
}

static int pack_raw_f32(sfd2_ctx *c, const TMap &m, ConvW &L, const std::string &conv, const std::string &bn, int cout,
                        size_t numel)
{
    const TView *w = find_t(m, conv + ".weight");
    if (!w || w->numel() != numel) return fail("missing or mis-shaped tensor: " + conv + ".weight");
    L.cout = cout; L.cout_pad = cout;
    std::vector<float> sc, sh;
    if (fold_scale_shift(m, conv, bn, cout, cout, sc, sh)) return -1;
    if (upload(L.w, w->d, numel * sizeof(float), c->stream)) return -1;
    if (set_scale_shift(c, L, sc, sh)) return -1;
    return 0;
}

static int pack_all_f32(sfd2_ctx *c, const TMap &m)
{
    if (pack_raw_f32(c, m, c->f1a, "conv1a.0", "conv1a.1", 64, 64 * 27)) return -1;
    if (pack_igemm_f32(c, m, c->f1b, "conv1b.0", "bn1b.0", 64, 64, 3, 2)) return -1;
    if (pack_igemm_f32(c, m, c->f2a, "conv2a.0", "conv2a.1", 64, 128, 3, 1)) return -1;
    if (pack_igemm_f32(c, m, c->f2b, "conv2b.0", "bn2b.0", 128, 128, 3, 2)) return -1;
    if (pack_igemm_f32(c, m, c->f3a, "conv3a.0", "conv3a.1", 128, 256, 3, 1)) return -1;
    if (pack_igemm_f32(c, m, c->f3b, "conv3b.0", "bn3b.0", 256, 256, 3, 1)) return -1;
    for (int b = 0; b < 3; ++b) {
        const std::string p = "conv4." + std::to_string(b) + ".";
        if (pack_igemm_f32(c, m, c->frb1[b], p + "conv1", p + "bn1", 256, 256, 1, 1)) return -1;
        if (pack_raw_f32(c, m, c->frb2[b], p + "conv2", p + "bn2", 256, 256 * 72)) return -1;
        if (pack_igemm_f32(c, m, c->frb3[b], p + "conv3", p + "bn3", 256, 256, 1, 1)) return -1;
    }
    if (pack_igemm_f32(c, m, c->fpa0, "convPa.0", "convPa.1", 256, 256, 3, 2)) return -1;
    if (pack_igemm_f32(c, m, c->fpa3, "convPa.3", "", 256, 256, 3, 1)) return -1;
    if (pack_igemm_f32(c, m, c->fda0, "convDa.0", "convDa.1", 256, 256, 3, 1)) return -1;
    if (pack_igemm_f32(c, m, c->fda3, "convDa.3", "", 256, 256, 3, 1)) return -1;
    if (pack_igemm_f32(c, m, c->fpb, "convPb", "", 256, 65, 1, 1)) return -1;
    if (pack_igemm_f32(c, m, c->fdb, "convDb", "", 256, 128, 1, 1)) return -1;
    return 0;
}

static int calibrate_on_probe(sfd2_ctx *c);

extern "C" int sfd2_load_weights(sfd2_ctx *c, const sfd2_tensor *tensors, int n)
{
    if (!c || !tensors) return fail("sfd2_load_weights: null argument");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    c->weights_loaded = false;   // a failure part-way must not leave a half-replaced set usable
    graphs_release(c);           // captured units hold pointers into the filter buffers
    {   // SFD2_PREC_F16X3 keeps split copies of the fp32 filters, made on first use: they belong to the OLD weights
        ConvW *fl[] = {&c->f1a, &c->f1b, &c->f2a, &c->f2b, &c->f3a, &c->f3b, &c->frb1[0], &c->frb1[1], &c->frb1[2], &c->frb2[0],
                       &c->frb2[1], &c->frb2[2], &c->frb3[0], &c->frb3[1], &c->frb3[2], &c->fpa0, &c->fpa3, &c->fda0, &c->fda3,
                       &c->fpb, &c->fdb};
        for (ConvW *L : fl) { L->wx3.release(); L->wx3p.release(); L->wsl.release(); }
    }
    TMap m;
    for (int i = 0; i < n; ++i) {
        if (!tensors[i].name || !tensors[i].data) continue;
        TView v;
        v.d = tensors[i].data;
        for (int d = 0; d < tensors[i].ndim && d < 4; ++d) v.shape.push_back(tensors[i].shape[d]);
        m[tensors[i].name] = v;
    }
    std::vector<std::vector<float>> owned;     // the normalised filter copies m points into from here on
    owned.reserve(32);
    {
        static const char *convs[] = {"conv1a.0", "conv1b.0", "conv2a.0", "conv2b.0", "conv3a.0", "conv3b.0", "convPa.0", "convPa.3",
                                      "convDa.0", "convDa.3", "convPb", "convDb"};
        for (const char *n : convs) normalise_filters(m, n, owned);
        for (int b = 0; b < 3; ++b)
            for (int i = 1; i <= 3; ++i) normalise_filters(m, "conv4." + std::to_string(b) + ".conv" + std::to_string(i), owned);
    }
    if (pack_conv1a(c, m)) return -1;
    if (pack_igemm(c, m, c->c1b, "conv1b.0", "bn1b.0", 64, 64, 3, 2)) return -1;
    {   // the same filters as [tap][oc][ic] for the fused stem kernel
        const TView *w = find_t(m, "conv1b.0.weight");
        std::vector<half_t> pk((size_t)9 * 64 * 64);
        for (int t = 0; t < 9; ++t)
            for (int oc = 0; oc < 64; ++oc)
                for (int ic = 0; ic < 64; ++ic) pk[((size_t)t * 64 + oc) * 64 + ic] = (half_t)w->d[((size_t)oc * 64 + ic) * 9 + t];
        if (upload(c->w1b_fused, pk.data(), pk.size() * sizeof(half_t), c->stream)) return -1;
    }
    {   // ... and as register fragments of the compensated fused stem (fused_stem_c_kernel.hip): [channel half][unit = tap * 2 +
        // half of the input channels][lane][K slice 0 | K slice 1 | corr fragment]
        const TView *w = find_t(m, "conv1b.0.weight");
        const int b0 = 127 - SFD2_C_XL_SHIFT - c->c1b.sbyte;
        std::vector<unsigned short> pk((size_t)2 * 18 * 64 * 32, 0);
        for (int cth = 0; cth < 2; ++cth)
            for (int u = 0; u < 18; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int kk = 0; kk < 2; ++kk)
                        for (int j = 0; j < 8; ++j) {
                            const int oc = cth * 32 + (lane & 31), ic = (u & 1) * 32 + kk * 16 + (lane >> 5) * 8 + j, tap = u >> 1;
                            const float v = w->d[((size_t)oc * 64 + ic) * 9 + tap];
                            const half_t h = (half_t)v;
                            unsigned short hb;
                            std::memcpy(&hb, &h, 2);
                            const size_t base = ((size_t)(cth * 18 + u) * 64 + lane) * 32;
                            pk[base + kk * 8 + j] = hb;
                            pk[base + 16 + kk * 8 + j] = (unsigned short)(f32_to_e4m3(std::ldexp(v, b0)) | (f32_to_e4m3(std::ldexp(v - (float)h, b0 + 11)) << 8));
                        }
        if (upload(c->w1b_stem_c, pk.data(), pk.size() * 2, c->stream)) return -1;
        {   // option "fp6_acts": the corr fragment (bytes 32 .. 63 of a lane's 64) = 32 e2m3 codes in the order of the fp6 half-records the
            // kernel's phase 1 writes (position p: j = p / 2 <-> input channel 8 (j / 4) + 4 (lane / 32) + j % 4 of the unit's 32; p even: w,
            // p odd: (w - fp16(w)) * 2^11), 24 bytes, then in dword 6 the output channel's E8M0 scale byte 2^(ec - 11), ec: max|w| / 2^ec <= 7.5
            std::vector<unsigned short> p6 = pk;
            for (int cth = 0; cth < 2; ++cth)
                for (int lane = 0; lane < 64; ++lane) {
                    const int oc = cth * 32 + (lane & 31), h = lane >> 5;
                    float mx = 0.0f;
                    for (int i = 0; i < 64 * 9; ++i) mx = std::max(mx, std::fabs(w->d[(size_t)oc * 576 + i]));
                    int ec = (mx > 0.0f && std::isfinite(mx)) ? (int)std::ceil(std::log2(mx / 7.5f)) : 0;
                    while (std::ldexp(mx, -ec) > 7.5f) ++ec;
                    ec = std::max(-40, std::min(40, ec));
                    for (int u = 0; u < 18; ++u) {
                        unsigned char str[32] = {0};
                        for (int p = 0; p < 32; ++p) {
                            const int j = p >> 1, ic = (u & 1) * 32 + 8 * (j >> 2) + 4 * h + (j & 3), tap = u >> 1;
                            const float v = w->d[((size_t)oc * 64 + ic) * 9 + tap];
                            const float x = (p & 1) ? std::ldexp(v - (float)(half_t)v, 11 - ec) : std::ldexp(v, -ec);
                            const unsigned int code = f32_to_e2m3(x);
                            const int bit = 6 * p;
                            str[bit >> 3] |= (unsigned char)(code << (bit & 7));
                            if ((bit & 7) > 2) str[(bit >> 3) + 1] |= (unsigned char)(code >> (8 - (bit & 7)));
                        }
                        const unsigned int sb = ((unsigned int)(127 - 11 + ec) & 255u) * 0x01010101u;
                        std::memcpy(str + 24, &sb, 4);
                        std::memcpy(reinterpret_cast<unsigned char *>(p6.data()) + (((size_t)(cth * 18 + u) * 64 + lane) * 32 + 16) * 2, str, 32);
                    }
                }
            if (upload(c->w1b_stem_c6, p6.data(), p6.size() * 2, c->stream)) return -1;
        }
        for (int cth = 0; cth < 2; ++cth)          // the same fragments with the lo' parts as fp16 (SFD2_PREC_F16X3)
            for (int u = 0; u < 18; ++u)
                for (int lane = 0; lane < 64; ++lane)
                    for (int kk = 0; kk < 2; ++kk)
                        for (int j = 0; j < 8; ++j) {
                            const int oc = cth * 32 + (lane & 31), ic = (u & 1) * 32 + kk * 16 + (lane >> 5) * 8 + j, tap = u >> 1;
                            const float v = w->d[((size_t)oc * 64 + ic) * 9 + tap];
                            const half_t l = (half_t)((v - (float)(half_t)v) * 2048.0f);
                            unsigned short lb;
                            std::memcpy(&lb, &l, 2);
                            pk[((size_t)(cth * 18 + u) * 64 + lane) * 32 + 16 + kk * 8 + j] = lb;
                        }
        if (upload(c->w1b_stem_x3, pk.data(), pk.size() * 2, c->stream)) return -1;
    }
    if (pack_igemm(c, m, c->c2a, "conv2a.0", "conv2a.1", 64, 128, 3, 1)) return -1;
    if (pack_igemm(c, m, c->c2b, "conv2b.0", "bn2b.0", 128, 128, 3, 2)) return -1;
    if (pack_igemm(c, m, c->c3a, "conv3a.0", "conv3a.1", 128, 256, 3, 1)) return -1;
    if (pack_igemm(c, m, c->c3b, "conv3b.0", "bn3b.0", 256, 256, 3, 1)) return -1;
    for (int b = 0; b < 3; ++b) {
        const std::string p = "conv4." + std::to_string(b) + ".";
        if (pack_igemm(c, m, c->rb1[b], p + "conv1", p + "bn1", 256, 256, 1, 1)) return -1;
        if (pack_gconv(c, m, c->rb2[b], p + "conv2", p + "bn2")) return -1;
        if (pack_igemm(c, m, c->rb3[b], p + "conv3", p + "bn3", 256, 256, 1, 1)) return -1;
    }
    if (pack_igemm(c, m, c->pa0, "convPa.0", "convPa.1", 256, 256, 3, 2)) return -1;
    if (pack_igemm(c, m, c->pa3, "convPa.3", "", 256, 256, 3, 1)) return -1;
    if (pack_igemm(c, m, c->da0, "convDa.0", "convDa.1", 256, 256, 3, 1)) return -1;
    if (pack_igemm(c, m, c->da3, "convDa.3", "", 256, 256, 3, 1)) return -1;
    HIPCHECK(c->da3.wsl.ensure((size_t)(256 / 32) * 9 * c->da3.cout_pad * 32 * sizeof(half_t)));     // the same filters in sparse_da3_kernel's fragment order
    launch_sparse_da3_repack(c->stream, c->da3.w.as<half_t>(), c->da3.wsl.as<half_t>(), c->da3.cout_pad, 256);
    if (pack_igemm(c, m, c->pb, "convPb", "", 256, 65, 1, 1)) return -1;
    if (pack_igemm(c, m, c->db, "convDb", "", 256, 128, 1, 1)) return -1;
    // ConvSta exists only in models built with require_stability=True (nets/sfd2.py:302-303); the reference loads
    // checkpoints without it (strict=False, extract_localization.py:214), so it is optional here and stability
    // requests are refused only when it is absent
    const TView *sw = find_t(m, "ConvSta.weight"), *sb = find_t(m, "ConvSta.bias");
    c->has_sta = false;
    if (sw || sb) {
        if (!sw || !sb) return fail("missing tensor: ConvSta.{weight,bias} (only one of the two is present)");
        if (sw->numel() != 3 * 256 || sb->numel() != 3) return fail("bad shape for ConvSta");
        c->h_sta_w.assign(sw->d, sw->d + 3 * 256);
        if (upload(c->sta_w, sw->d, 3 * 256 * sizeof(float), c->stream)) return -1;
        if (upload(c->sta_b, sb->d, 3 * sizeof(float), c->stream)) return -1;
        c->has_sta = true;
    }
    if (pack_all_f32(c, m)) return -1;
    for (int g = 0; g < AE_COUNT; ++g) { c->act_exp[g] = 0; c->act_max[g] = 0.0f; }
    c->weights_loaded = true;
    if (apply_act_exponents(c)) { c->weights_loaded = false; return -1; }
    c->margin_done = false;        // the self-check's findings belong to the weights it ran on
    c->margin_pending = false;
    c->margin_choice = -1;
    for (float &e : c->margin_err) e = -1.0f;
    c->relax_err = -1.0f;
    c->opt_c3b_plain = c->user_c3b_plain == 1 ? 1 : 0;     // (never on unverified: -1 waits for the self-check)
    if (c->opt_auto_range && calibrate_on_probe(c)) { c->weights_loaded = false; return -1; }
    return 0;
}

// ------------------------------------------------------------------------------------------ activation exponents
int apply_act_exponents(sfd2_ctx *c)
{
    struct Row { ConvW *L; int e_in, e_out; };
    const int *e = c->act_exp;
    std::vector<Row> rows = {
        {&c->c1a, -1, AE_CONV1A}, {&c->c1b, AE_CONV1A, AE_CONV1B}, {&c->c2a, AE_CONV1B, AE_CONV2A}, {&c->c2b, AE_CONV2A, AE_CONV2B},
        {&c->c3a, AE_CONV2B, AE_CONV3A}, {&c->c3b, AE_CONV3A, AE_TRUNK}, {&c->pa0, AE_TRUNK, AE_PA0}, {&c->pa3, AE_PA0, -1},
        {&c->da0, AE_TRUNK, AE_DA0}, {&c->da3, AE_DA0, -1}};
    for (int b = 0; b < 3; ++b) {
        rows.push_back({&c->rb1[b], AE_TRUNK, AE_T1_0 + b});
        rows.push_back({&c->rb2[b], AE_T1_0 + b, AE_T2_0 + b});
        rows.push_back({&c->rb3[b], AE_T2_0 + b, AE_TRUNK});
    }
    std::vector<float> sc, sh;
    for (const Row &r : rows) {
        ConvW &L = *r.L;
        if (L.h_scale.empty()) continue;
        const int ei = r.e_in >= 0 ? e[r.e_in] : 0, eo = r.e_out >= 0 ? e[r.e_out] : 0;
        sc = L.h_scale; sh = L.h_shift;
        for (float &v : sc) v = std::ldexp(v, eo - ei);
        for (float &v : sh) v = std::ldexp(v, eo);
        if (upload(L.scale, sc.data(), sc.size() * sizeof(float), c->stream)) return -1;
        if (upload(L.shift, sh.data(), sh.size() * sizeof(float), c->stream)) return -1;
        if (!L.h_sa6.empty() && L.sa6.p) {          // conv3x3_pp's fp6 form keeps its own copy of the shifts
            std::vector<int> sa = L.h_sa6;
            std::memcpy(sa.data(), sh.data(), std::min(sh.size(), sa.size() / 2) * sizeof(float));
            if (upload(L.sa6, sa.data(), sa.size() * sizeof(int), c->stream)) return -1;
        }
        if (!L.h_sa66.empty() && L.sa66.p) {
            std::vector<int> sa = L.h_sa66;
            std::memcpy(sa.data(), sh.data(), std::min(sh.size(), sa.size() / 2) * sizeof(float));
            if (upload(L.sa66, sa.data(), sa.size() * sizeof(int), c->stream)) return -1;
        }
    }
    if (!c->da0.h_scale.empty()) {                  // option "x3_desc16": convDa.0 in fp16 over an input that is NOT scaled by 2^e[AE_TRUNK]
        sc = c->da0.h_scale;
        for (float &v : sc) v = std::ldexp(v, e[AE_DA0]);
        if (upload(c->da0.scale_rawin, sc.data(), sc.size() * sizeof(float), c->stream)) return -1;
    }
    if (c->has_sta && !c->h_sta_w.empty()) {        // ConvSta reads the backbone output with fp32 filters of its own
        std::vector<float> w = c->h_sta_w;
        for (float &v : w) v = std::ldexp(v, -e[AE_TRUNK]);
        if (upload(c->sta_w16, w.data(), w.size() * sizeof(float), c->stream)) return -1;
    }
    graphs_release(c);      // (captured units hold nothing of this by value, but a unit captured mid-way must not survive a change of scale)
    return reset_range_records(c);   // maxima recorded under the previous exponents would mix two scalings in sfd2_get_range_status
}

// Range calibration.  The image goes through the network ONCE in SFD2_PREC_F32 on the parity entry point (every activation in its
// own fp32 buffer, at the network's own scale), the largest |x| of every stored tensor is reduced on the device, and each group's
// exponent becomes the power of two that brings that maximum to SFD2_RANGE_TARGET.  One pass: the fp32 tensors do not depend on
// the exponents.  The CPU study of the compensated mode (tools/conditioning_sweep.py) has descriptors within
// 5e-4 for tensor maxima from 0.2 to 500, 1.1e-3 at 0.03 and a cliff above 1792: 16 sits 2^7 below the cliff and 2^9 above the floor.
#define SFD2_RANGE_TARGET 16.0f
static int calibrate_impl(sfd2_ctx *c, const float *img, int on_device, int H, int W, int flags)
{
    if (!c->weights_loaded) return fail("sfd2_calibrate_range: weights not loaded");
    HIPCHECK(hipSetDevice(c->device));
    const int prec = c->precision, prof_max = c->prof_max_steps;
    c->precision = SFD2_PREC_F32;
    c->prof_max_steps = 0;      // the calibration pass is not one of the caller's profiled steps
    const int rc = sfd2_det(c, img, on_device, H, W, flags, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    c->precision = prec;
    c->prof_max_steps = prof_max;
    if (rc) return -1;
    HIPCHECK(c->range_scratch.ensure(AE_COUNT * sizeof(unsigned int)));
    HIPCHECK(hipMemsetAsync(c->range_scratch.p, 0, AE_COUNT * sizeof(unsigned int), c->stream));
    static const struct { const char *name; int group; } tens[] = {
        {"conv1a", AE_CONV1A}, {"bn1b", AE_CONV1B}, {"conv2a", AE_CONV2A}, {"bn2b", AE_CONV2B}, {"conv3a", AE_CONV3A},
        {"bn3b", AE_TRUNK}, {"conv4.0", AE_TRUNK}, {"conv4.1", AE_TRUNK}, {"conv4.2", AE_TRUNK},
        {"conv4.0.bn1", AE_T1_0}, {"conv4.1.bn1", AE_T1_1}, {"conv4.2.bn1", AE_T1_2},
        {"conv4.0.bn2", AE_T2_0}, {"conv4.1.bn2", AE_T2_1}, {"conv4.2.bn2", AE_T2_2}, {"convPa.0", AE_PA0}, {"convDa.0", AE_DA0}};
    for (const auto &t : tens) {
        auto it = c->acts.find(t.name);
        if (it == c->acts.end() || !it->second.f32 || it->second.planar || it->second.pitch != it->second.c)
            return fail(std::string("sfd2_calibrate_range: activation not available: ") + t.name);
        launch_absmax_f32(c->stream, reinterpret_cast<const float *>(it->second.p), (size_t)it->second.c * it->second.h * it->second.w,
                          c->range_scratch.as<unsigned int>() + t.group);
    }
    HIPCHECK(hipGetLastError());
    float mx[AE_COUNT];
    HIPCHECK(hipMemcpyAsync(mx, c->range_scratch.p, sizeof(mx), hipMemcpyDeviceToHost, c->stream));
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int g = 0; g < AE_COUNT; ++g) {
        c->act_max[g] = mx[g];
        int e = 0;
        if (mx[g] > 0.0f && std::isfinite(mx[g])) e = (int)std::lround(std::log2(SFD2_RANGE_TARGET / mx[g]));
        c->act_exp[g] = std::max(-60, std::min(60, e));
    }
    return apply_act_exponents(c);
}

extern "C" int sfd2_calibrate_range(sfd2_ctx *c, const float *img, int img_on_device, int H, int W, int flags)
{
    if (!c || !img) return fail("sfd2_calibrate_range: null argument");
    return calibrate_impl(c, img, img_on_device, H, W, flags);
}

extern "C" int sfd2_get_act_exponents(sfd2_ctx *c, int32_t *exps, float *maxima, int cap, int *n)
{
    if (!c || !n) return fail("sfd2_get_act_exponents: null argument");
    *n = AE_COUNT;
    for (int g = 0; g < AE_COUNT && g < cap; ++g) {
        if (exps) exps[g] = c->act_exp[g];
        if (maxima) maxima[g] = c->act_max[g];
    }
    return 0;
}

extern "C" int sfd2_set_act_exponents(sfd2_ctx *c, const int32_t *exps, int n)
{
    if (!c || (n > 0 && !exps)) return fail("sfd2_set_act_exponents: null argument");
    if (!c->weights_loaded) return fail("sfd2_set_act_exponents: weights not loaded");
    if (n != 0 && n != AE_COUNT) return fail("sfd2_set_act_exponents: expected " + std::to_string((int)AE_COUNT) + " exponents (or 0 to clear)");
    HIPCHECK(hipSetDevice(c->device));
    HIPCHECK(hipStreamSynchronize(c->stream));
    for (int g = 0; g < AE_COUNT; ++g) {
        const int e = n ? exps[g] : 0;
        if (e < -60 || e > 60) return fail("sfd2_set_act_exponents: exponent out of range");
        c->act_exp[g] = e;
    }
    return apply_act_exponents(c);
}

// Self-check of SFD2_PREC_F16C (option "auto_margin", default 1; VERDICT r4 weak #3: the tolerance margin on heavy-tailed weights is thin).  How far the
// compensated mode's descriptors sit from the fp32 reference depends on the CHECKPOINT: 4.5-5.3e-4 on the dense map of the probe for the well-conditioned
// draw, 8-10e-4 for Student-t filters, 6.5-8.2e-4 for filters that do not sum to zero (profiles/r05j_probe_err.txt) -- and nobody has run the real
// checkpoint (/root/reference/.MISSING_LARGE_BLOBS).  So the library measures it: the probe goes through sfd2_det once more in SFD2_PREC_F32 and in
// SFD2_PREC_F16C, the L2-normalised descriptor maps are compared, and while the largest difference exceeds SFD2_MARGIN_TARGET the accuracy options are
// tried in order of their cost -- "rb_inner" = 0 (the tensors inside the ResBlocks compensated too: +6 % per extract), "comp_heads" = 1 (+24 %), both
// (+30 %) -- and the first one that meets the target stays on.  sfd2_get_margin_status reports the four errors and the choice; an option the caller sets
// afterwards overrides it.  The dense map's maximum is a stricter measure than the error at key points (more samples, flat regions included).
#define SFD2_MARGIN_TARGET 7.0e-4f
// "c3b_plain" spends margin for speed, so it is held to a tighter figure than the one that buys margin back: over eight seeds per weight family
// (profiles/r06p_relax_probe_8seeds.txt) an extraction's descriptor error is up to 1.1x the probe's with the option on (probe 6.59e-4 -> 7.28e-4 at 480x640, the worst of 18
// cases that had it on under a 7.0e-4 limit); with 6.5e-4 the worst case that keeps it is 6.9e-4.
#define SFD2_RELAX_TARGET 6.5e-4f
static int probe_descriptors(sfd2_ctx *c, const float *img, int H, int W, int prec, std::vector<float> &out)
{
    const int p0 = c->precision, pm = c->prof_max_steps;
    c->precision = prec;
    c->prof_max_steps = 0;
    int hc = 0, wc = 0;
    out.assign((size_t)128 * ((H + 3) / 4 + 1) * ((W + 3) / 4 + 1), 0.0f);
    const int rc = sfd2_det(c, img, 0, H, W, 0, nullptr, nullptr, out.data(), 0, nullptr, nullptr, &hc, &wc);
    c->precision = p0;
    c->prof_max_steps = pm;
    if (rc) return -1;
    out.resize((size_t)128 * hc * wc);
    return 0;
}

static int margin_selfcheck(sfd2_ctx *c, const float *img, int H, int W)
{
    std::vector<float> ref, got;
    if (probe_descriptors(c, img, H, W, SFD2_PREC_F32, ref)) return -1;
    const int rb0 = c->user_rb_inner, ch0 = c->user_comp_heads;      // (not what an earlier load's self-check left behind)
    // "c3b_plain" (round 6): -1 = try it and keep it when the probe stays inside the target with it, 1 = the caller wants it (every run below has it), 0 = off
    const int forced_plain = c->user_c3b_plain == 1 ? 1 : 0;
    auto run = [&](int plain, int rbi, int chd, float &err) -> int {
        c->opt_c3b_plain = plain;
        c->opt_rb_inner = rbi;
        c->opt_comp_heads = chd;
        if (probe_descriptors(c, img, H, W, SFD2_PREC_F16C, got)) return -1;
        if (got.size() != ref.size()) return fail("auto_margin: descriptor maps of different size");
        float m = 0.0f;
        for (size_t i = 0; i < ref.size(); ++i) {
            const float d = std::fabs(got[i] - ref[i]);
            m = (d > m || !(d == d)) ? (d == d ? d : INFINITY) : m;      // (a NaN counts as infinitely wrong)
        }
        err = m;
        return 0;
    };
    for (float &e : c->margin_err) e = -1.0f;
    c->relax_err = -1.0f;
    c->margin_choice = 0;
    int rc = 0;
    if (c->user_c3b_plain == -1) rc = run(1, rb0, ch0, c->relax_err);
    if (rc == 0) rc = run(forced_plain, rb0, ch0, c->margin_err[0]);
    if (forced_plain) c->relax_err = c->margin_err[0];
    // the accuracy options in order of their cost; a key the caller set explicitly is reported, not overridden
    const bool may_rb = !c->user_set_rb_inner && rb0 != 0, may_ch = !c->user_set_comp_heads && ch0 != 1;
    if (rc == 0 && c->margin_err[0] > SFD2_MARGIN_TARGET) {
        bool ok = false;
        if (may_rb) { rc = run(forced_plain, 0, ch0, c->margin_err[1]); c->margin_choice = 1; ok = rc == 0 && c->margin_err[1] <= SFD2_MARGIN_TARGET; }
        if (rc == 0 && !ok && may_ch) { rc = run(forced_plain, rb0, 1, c->margin_err[2]); c->margin_choice = 2; ok = rc == 0 && c->margin_err[2] <= SFD2_MARGIN_TARGET; }
        if (rc == 0 && !ok && may_rb && may_ch) { rc = run(forced_plain, 0, 1, c->margin_err[3]); c->margin_choice = 3; }
    }
    if (rc) { c->opt_rb_inner = rb0; c->opt_comp_heads = ch0; c->opt_c3b_plain = forced_plain; c->margin_choice = -1; return -1; }
    c->opt_rb_inner = (c->margin_choice & 1) ? 0 : rb0;
    c->opt_comp_heads = (c->margin_choice & 2) ? 1 : ch0;
    // conv3b without its correction chunks only where the probe says the checkpoint has the room: the options as set are inside the target AND stay inside with it
    c->opt_c3b_plain = (forced_plain || (c->user_c3b_plain == -1 && c->margin_choice == 0 && c->relax_err >= 0.0f && c->relax_err <= SFD2_RELAX_TARGET)) ? 1 : 0;
    c->margin_done = true;
    static const bool verbose = sfd2_env("SFD2_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "sfd2: f16c self-check: probe error %.2e (target %.1e; with c3b_plain %.2e, its limit %.1e) -> rb_inner %d, comp_heads %d, c3b_plain %d\n",
                c->margin_err[0], SFD2_MARGIN_TARGET, c->relax_err, SFD2_RELAX_TARGET, c->opt_rb_inner, c->opt_comp_heads, c->opt_c3b_plain);
    graphs_release(c);
    return reset_range_records(c);      // (the probe's maxima are not the caller's images')
}

extern "C" int sfd2_get_margin_status(sfd2_ctx *c, float *errs4, int *choice, float *target)
{
    if (!c) return fail("sfd2_get_margin_status: null argument");
    if (errs4) for (int i = 0; i < 4; ++i) errs4[i] = c->margin_err[i];
    if (choice) *choice = c->margin_choice;
    if (target) *target = SFD2_MARGIN_TARGET;
    return 0;
}

extern "C" int sfd2_get_relax_status(sfd2_ctx *c, float *err_plain, int *c3b_plain)
{
    if (!c) return fail("sfd2_get_relax_status: null argument");
    if (err_plain) *err_plain = c->relax_err;
    if (c3b_plain) *c3b_plain = c->opt_c3b_plain;
    return 0;
}

// The built-in probe: 192 x 256, half white noise and half a blocky low-frequency field (like the synthetic images of the tests), from
// a fixed xorshift stream -- what sfd2_load_weights calibrates on when nothing better has been shown to the context yet.  A network
// with BatchNorm after every conv keeps its activations at the same order of magnitude on any image; the envelope is 2^12 wide.
static void probe_image(std::vector<float> &img, int &H, int &W)
{
    H = 192; W = 256;
    img.resize((size_t)3 * H * W);
    unsigned int s = 0x9E3779B9u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    std::vector<float> coarse((size_t)3 * (H / 16) * (W / 16));
    for (float &v : coarse) v = rnd();
    for (int ch = 0; ch < 3; ++ch)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                img[((size_t)ch * H + y) * W + x] = 0.5f * rnd() + 0.5f * coarse[((size_t)ch * (H / 16) + y / 16) * (W / 16) + x / 16];
}

static void release_probe_workspace(sfd2_ctx *c)
{
    // the probe's fp32 parity workspace (every activation in its own fp32 buffer) is of no use to a throughput context: give it back
    // (a context that runs SFD2_PREC_F32 / F16X3 allocates what its own geometry needs on its first call)
    DevBuf *f32ws[] = {&c->g1a, &c->g1b, &c->g2a, &c->g2b, &c->g3a, &c->g3b, &c->gpa0_o, &c->gpa_o, &c->gda0_o, &c->gda_o};
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : f32ws) b->release();
    for (int b = 0; b < 3; ++b) { c->grt1[b].release(); c->grt2[b].release(); c->gro[b].release(); }
    c->acts.clear();            // (they named those buffers)
}

// The self-check belongs to SFD2_PREC_F16C: a context in another precision does not pay for it at load -- it runs when the
// precision is switched to F16C later (sfd2_set_precision), on the same probe
int sfd2_margin_selfcheck_if_pending(sfd2_ctx *c)
{
    if (!c->weights_loaded || !c->opt_auto_margin || !c->margin_pending || c->precision != SFD2_PREC_F16C) return 0;
    c->margin_pending = false;
    std::vector<float> img;
    int H, W;
    probe_image(img, H, W);
    const int rc = margin_selfcheck(c, img.data(), H, W);
    release_probe_workspace(c);
    return rc;
}

static int calibrate_on_probe(sfd2_ctx *c)
{
    int H, W;
    std::vector<float> img;
    probe_image(img, H, W);
    int rc = calibrate_impl(c, img.data(), 0, H, W, 0);
    if (rc == 0 && c->opt_auto_margin) {
        if (c->precision == SFD2_PREC_F16C) rc = margin_selfcheck(c, img.data(), H, W);
        else c->margin_pending = true;
    }
    release_probe_workspace(c);
    return rc;
}
