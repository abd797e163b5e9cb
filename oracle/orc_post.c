/*
 * ORACLE (test infrastructure, NOT product code) -- CPU restatement of the
 * non-GEMM arithmetic on the SFD2 hot path: detector / descriptor / stability
 * heads, NMS, key-point selection, descriptor sampling and the nearest-neighbour
 * matchers.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library.  Built with -ffp-contract=off: every float expression
 * below is evaluated exactly as written (one rounding per operation).
 *
 * Parity pinning: checked against the .npz files under tests/golden (produced by importing the
 * reference in the authoring container, tests/golden/gen_goldens.py).  The
 * reference ships no tests of its own for this path.
 *
 * Tie rule (the reference's own order among equal scores is implementation
 * defined: numpy introsort at nets/extractor.py:176,323; torch.topk):
 *   key-points: score descending, then row-major pixel index ascending;
 *   matcher arg-max / top-2: lowest index first.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- nets/sfd2.py:329-337 : exp, soft-max style normalisation (no max
 * subtraction, +1e-5 in the denominator), drop the dust-bin channel 64 and
 * depth-to-space by 8.  logits[65][hc][wc] -> score[8*hc][8*wc] */
void orc_detector_head(const float *logits, int hc, int wc, float *score)
{
    const size_t plane = (size_t)hc * wc;
    for (int y = 0; y < hc; ++y)
        for (int x = 0; x < wc; ++x) {
            float e[65];
            float s = 0.0f;
            for (int c = 0; c < 65; ++c) {
                e[c] = expf(logits[c * plane + (size_t)y * wc + x]);
                s += e[c];
            }
            const float den = s + 0.00001f;
            for (int c = 0; c < 64; ++c) {
                const int i = c >> 3, j = c & 7;
                score[(size_t)(8 * y + i) * (8 * wc) + 8 * x + j] = e[c] / den;
            }
        }
}

/* ---- nets/sfd2.py:342 : F.normalize(desc, dim=1): x / max(||x||_2, 1e-12) over C */
void orc_l2norm_channels(float *x, int c, size_t hw)
{
    for (size_t p = 0; p < hw; ++p) {
        float s = 0.0f;
        for (int ch = 0; ch < c; ++ch) {
            const float v = x[(size_t)ch * hw + p];
            s += v * v;
        }
        float n = sqrtf(s);
        if (n < 1e-12f) n = 1e-12f;
        for (int ch = 0; ch < c; ++ch) x[(size_t)ch * hw + p] /= n;
    }
}

/* ---- F.interpolate(mode='bilinear', align_corners=False), size given
 * (nets/sfd2.py:346, nets/extractor.py:138).  torch semantics:
 *   scale = in / out (float);  src = scale * (dst + 0.5) - 0.5, clamped to >= 0
 *   i0 = min(floor(src), in-1); i1 = min(i0 + 1, in - 1); l1 = clamp(src - i0, 0, 1); l0 = 1 - l1
 *   out = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11), with the
 *   rounding sequence given in orc_resize_bilinear below.                         */
static void lin_coeff(int dst, int in_size, int out_size, int *i0, int *i1, float *l0, float *l1)
{
    if (in_size == out_size) {
        *i0 = dst; *i1 = dst; *l0 = 1.0f; *l1 = 0.0f;
        return;
    }
    const float scale = (float)in_size / (float)out_size;
    float src = fmaf(scale, (float)dst + 0.5f, -0.5f);  /* single rounding, as torch's build contracts it */
    if (src < 0.0f) src = 0.0f;
    int a = (int)floorf(src);
    if (a > in_size - 1) a = in_size - 1;
    float lam = src - (float)a;
    if (lam < 0.0f) lam = 0.0f;
    if (lam > 1.0f) lam = 1.0f;
    *i0 = a;
    *i1 = (a + 1 < in_size) ? a + 1 : in_size - 1;
    *l1 = lam;
    *l0 = 1.0f - lam;
}

void orc_resize_bilinear(const float *in, int c, int h, int w, int oh, int ow, float *out)
{
    for (int ch = 0; ch < c; ++ch)
        for (int y = 0; y < oh; ++y) {
            int y0, y1; float ly0, ly1;
            lin_coeff(y, h, oh, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < ow; ++x) {
                int x0, x1; float lx0, lx1;
                lin_coeff(x, w, ow, &x0, &x1, &lx0, &lx1);
                const float *p = in + (size_t)ch * h * w;
                const float v00 = p[(size_t)y0 * w + x0], v01 = p[(size_t)y0 * w + x1];
                const float v10 = p[(size_t)y1 * w + x0], v11 = p[(size_t)y1 * w + x1];
                /* Rounding sequence of torch's CPU kernel as built (FMA-contracted):
                 * fma(v0, w0, v1*w1) at both levels.  Found by enumerating the candidate
                 * orders against torch 2.10 (bit-exact on 25x33->100x130, 300x400->1200x1600,
                 * 104x136->100x130) and pinned by tests/test_oracle_vs_golden.py. */
                const float top = fmaf(v00, lx0, v01 * lx1);
                const float bot = fmaf(v10, lx0, v11 * lx1);
                out[((size_t)ch * oh + y) * ow + x] = fmaf(top, ly0, bot * ly1);
            }
        }
}

/* ---- nets/sfd2.py:305-311 : arg-max over the 3 stability classes (first maximum
 * wins, as torch.max does) -> {0: 0.1, 1: 0.5, 2: 1.0} */
void orc_cls_to_value(const float *x3, size_t hw, float *out)
{
    for (size_t p = 0; p < hw; ++p) {
        int best = 0;
        float bv = x3[p];
        for (int c = 1; c < 3; ++c)
            if (x3[(size_t)c * hw + p] > bv) { bv = x3[(size_t)c * hw + p]; best = c; }
        out[p] = best == 0 ? 0.1f : (best == 1 ? 0.5f : 1.0f);
    }
}

/* ---- nets/extractor.py:20-35 simple_nms, one [h][w] map.
 * max_pool2d(kernel 2r+1, stride 1, padding r) pads with -inf. */
static void maxpool(const float *in, int h, int w, int r, float *tmp, float *out)
{
    /* separable: rows then columns */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float m = -INFINITY;
            const int a = x - r < 0 ? 0 : x - r, b = x + r >= w ? w - 1 : x + r;
            for (int t = a; t <= b; ++t) if (in[(size_t)y * w + t] > m) m = in[(size_t)y * w + t];
            tmp[(size_t)y * w + x] = m;
        }
    for (int y = 0; y < h; ++y) {
        const int a = y - r < 0 ? 0 : y - r, b = y + r >= h ? h - 1 : y + r;
        for (int x = 0; x < w; ++x) {
            float m = -INFINITY;
            for (int t = a; t <= b; ++t) if (tmp[(size_t)t * w + x] > m) m = tmp[(size_t)t * w + x];
            out[(size_t)y * w + x] = m;
        }
    }
}

void orc_simple_nms(const float *scores, int h, int w, int radius, float *out)
{
    const size_t n = (size_t)h * w;
    float *tmp = (float *)malloc(n * sizeof(float));
    float *mp = (float *)malloc(n * sizeof(float));
    float *mask = (float *)malloc(n * sizeof(float));  /* max_mask as 0/1 float */
    float *supp = (float *)malloc(n * sizeof(float));
    float *ss = (float *)malloc(n * sizeof(float));
    maxpool(scores, h, w, radius, tmp, mp);
    for (size_t i = 0; i < n; ++i) mask[i] = scores[i] == mp[i] ? 1.0f : 0.0f;
    for (int it = 0; it < 2; ++it) {
        maxpool(mask, h, w, radius, tmp, supp);                 /* supp_mask = pool(mask) > 0 */
        for (size_t i = 0; i < n; ++i) ss[i] = supp[i] > 0.0f ? 0.0f : scores[i];
        maxpool(ss, h, w, radius, tmp, mp);
        for (size_t i = 0; i < n; ++i) {
            const int new_max = ss[i] == mp[i];
            if (new_max && !(supp[i] > 0.0f)) mask[i] = 1.0f;
        }
    }
    for (size_t i = 0; i < n; ++i) out[i] = mask[i] != 0.0f ? scores[i] : 0.0f;
    free(tmp); free(mp); free(mask); free(supp); free(ss);
}

/* ---- nets/extractor.py:158-183,322-326 : threshold (> conf_th), row-major
 * candidate list, sort by score descending, drop a `border`-pixel frame, keep the
 * top_k best (top_k <= 0: all).  Returns n; kpts_xy[n][2] (x, y), scores[n],
 * lin_idx[n] = y*w + x. */
typedef struct { float s; int64_t idx; } cand_t;
static int cand_cmp(const void *a, const void *b)
{
    const cand_t *p = (const cand_t *)a, *q = (const cand_t *)b;
    if (p->s > q->s) return -1;
    if (p->s < q->s) return 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

int64_t orc_select_keypoints(const float *nms, int h, int w, float conf_th, int border, int64_t top_k,
                             float *kpts_xy, float *scores, int64_t *lin_idx, int64_t cap)
{
    const size_t n = (size_t)h * w;
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) if (nms[i] > conf_th) ++cnt;
    cand_t *c = (cand_t *)malloc((cnt ? cnt : 1) * sizeof(cand_t));
    size_t k = 0;
    for (size_t i = 0; i < n; ++i)
        if (nms[i] > conf_th) {
            const int y = (int)(i / w), x = (int)(i % w);
            if (x < border || x >= w - border || y < border || y >= h - border) continue;
            c[k].s = nms[i]; c[k].idx = (int64_t)i; ++k;
        }
    qsort(c, k, sizeof(cand_t), cand_cmp);
    int64_t m = (int64_t)k;
    if (top_k > 0 && m > top_k) m = top_k;
    if (m > cap) m = cap;
    for (int64_t i = 0; i < m; ++i) {
        kpts_xy[2 * i] = (float)(c[i].idx % w);
        kpts_xy[2 * i + 1] = (float)(c[i].idx / w);
        scores[i] = c[i].s;
        lin_idx[i] = c[i].idx;
    }
    free(c);
    return m;
}

/* ---- nets/extractor.py:199-208 : grid coordinates gx = x/(nw/2) - 1 (fp32),
 * torch grid_sample(bilinear, zeros padding, align_corners=False):
 *   ix = ((gx + 1) * wc - 1) / 2, floor, 4 taps with out-of-range taps = 0,
 * then desc /= ||desc||_2 (np.linalg.norm, no epsilon).
 * desc_map[c][hc][wc] (already L2-normalised over c), kpts (x,y) -> out[n][c] */
void orc_sample_descriptors(const float *desc_map, int c, int hc, int wc, int nh, int nw,
                            const float *kpts_xy, int64_t n, float *out)
{
    const size_t plane = (size_t)hc * wc;
    for (int64_t i = 0; i < n; ++i) {
        const float gx = kpts_xy[2 * i] / ((float)nw / 2.0f) - 1.0f;
        const float gy = kpts_xy[2 * i + 1] / ((float)nh / 2.0f) - 1.0f;
        const float ix = ((gx + 1.0f) * (float)wc - 1.0f) / 2.0f;
        const float iy = ((gy + 1.0f) * (float)hc - 1.0f) / 2.0f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float w_nw = ((fx + 1.0f) - ix) * ((fy + 1.0f) - iy);
        const float w_ne = (ix - fx) * ((fy + 1.0f) - iy);
        const float w_sw = ((fx + 1.0f) - ix) * (iy - fy);
        const float w_se = (ix - fx) * (iy - fy);
        const int vx0 = x0 >= 0 && x0 < wc, vx1 = x1 >= 0 && x1 < wc;
        const int vy0 = y0 >= 0 && y0 < hc, vy1 = y1 >= 0 && y1 < hc;
        float ss = 0.0f;
        for (int ch = 0; ch < c; ++ch) {
            const float *p = desc_map + (size_t)ch * plane;
            float v = 0.0f;
            if (vy0 && vx0) v += p[(size_t)y0 * wc + x0] * w_nw;
            if (vy0 && vx1) v += p[(size_t)y0 * wc + x1] * w_ne;
            if (vy1 && vx0) v += p[(size_t)y1 * wc + x0] * w_sw;
            if (vy1 && vx1) v += p[(size_t)y1 * wc + x1] * w_se;
            out[(size_t)i * c + ch] = v;
            ss += v * v;
        }
        const float nrm = sqrtf(ss);
        for (int ch = 0; ch < c; ++ch) out[(size_t)i * c + ch] /= nrm;
    }
}

/* ---- hloc/matchers/nearest_neighbor.py:6-16 find_nn on one direction.
 * sim[n][m] (row stride ld_r, col stride ld_c so the transpose is free).
 * ratio_thresh <= 0 / dist_thresh <= 0 mean "None". */
void orc_find_nn(const float *sim, int n, int m, size_t ld_r, size_t ld_c,
                 float ratio_thresh, float dist_thresh, int64_t *matches, float *scores)
{
    for (int i = 0; i < n; ++i) {
        float b0 = -INFINITY, b1 = -INFINITY;
        int64_t i0 = -1;
        for (int j = 0; j < m; ++j) {
            const float v = sim[(size_t)i * ld_r + (size_t)j * ld_c];
            if (v > b0) { b1 = b0; b0 = v; i0 = j; }
            else if (v > b1) b1 = v;
        }
        const float d0 = 2.0f * (1.0f - b0), d1 = 2.0f * (1.0f - b1);
        int ok = 1;
        if (ratio_thresh > 0.0f) ok = ok && (d0 <= (ratio_thresh * ratio_thresh) * d1);
        if (dist_thresh > 0.0f) ok = ok && (d0 <= dist_thresh * dist_thresh);
        matches[i] = ok ? i0 : -1;
        scores[i] = ok ? (b0 + 1.0f) / 2.0f : 0.0f;
    }
}

/* ---- hloc/matchers/nearest_neighbor.py:19-24 mutual_check */
void orc_mutual_check(int64_t *m0, int n, const int64_t *m1)
{
    for (int i = 0; i < n; ++i)
        if (m0[i] > -1 && m1[m0[i]] != i) m0[i] = -1;
}

/* ---- it_loc/matcher.py:91-119 Matcher.forward with mode nnm (:122-130) or
 * nnr (:165-194, ratio = conf distance_threshold), fp64 similarities.
 * matches0[n] (-1 = none), scores0[n] = row maximum of sim (raw, never masked). */
void orc_itloc_match(const double *sim, int n, int m, int mode_nnr, double ratio,
                     int64_t *matches0, double *scores0)
{
    int64_t *nn12 = (int64_t *)malloc(n * sizeof(int64_t));
    int64_t *nn21 = (int64_t *)malloc(m * sizeof(int64_t));
    double *r12 = (double *)malloc(n * sizeof(double));
    double *r21 = (double *)malloc(m * sizeof(double));
    for (int i = 0; i < n; ++i) {
        double b0 = -INFINITY, b1 = -INFINITY; int64_t i0 = -1;
        for (int j = 0; j < m; ++j) {
            const double v = sim[(size_t)i * m + j];
            if (v > b0) { b1 = b0; b0 = v; i0 = j; } else if (v > b1) b1 = v;
        }
        nn12[i] = i0; scores0[i] = b0;
        r12[i] = sqrt(2.0 - 2.0 * b0) / (sqrt(2.0 - 2.0 * b1) + 1e-8);
    }
    for (int j = 0; j < m; ++j) {
        double b0 = -INFINITY, b1 = -INFINITY; int64_t i0 = -1;
        for (int i = 0; i < n; ++i) {
            const double v = sim[(size_t)i * m + j];
            if (v > b0) { b1 = b0; b0 = v; i0 = i; } else if (v > b1) b1 = v;
        }
        nn21[j] = i0;
        r21[j] = sqrt(2.0 - 2.0 * b0) / (sqrt(2.0 - 2.0 * b1) + 1e-8);
    }
    for (int i = 0; i < n; ++i) {
        int ok = nn21[nn12[i]] == i;
        if (mode_nnr) ok = ok && r12[i] <= ratio && r21[nn12[i]] <= ratio;
        matches0[i] = ok ? nn12[i] : -1;
    }
    free(nn12); free(nn21); free(r12); free(r21);
}

/* ---- extract.py:17-84 nms_fast: greedy grid NMS in score order.
 * cand (x, y, score) [n]; returns kept count; keep_order[] = indices into the
 * score-sorted candidate list, in the reference's output order (score descending).
 * Sort rule: score descending, then input index ascending (np.argsort(-score) with
 * a stable sort would give exactly that; numpy's default is not stable). */
typedef struct { float s; int64_t idx; } sc_t;
static int sc_cmp(const void *a, const void *b)
{
    const sc_t *p = (const sc_t *)a, *q = (const sc_t *)b;
    if (p->s > q->s) return -1;
    if (p->s < q->s) return 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

int64_t orc_nms_fast(const float *xs, const float *ys, const float *sc, int64_t n,
                     int h, int w, int dist, int64_t *keep /* indices into the input, score-desc order */)
{
    if (n == 0) return 0;
    sc_t *o = (sc_t *)malloc(n * sizeof(sc_t));
    for (int64_t i = 0; i < n; ++i) { o[i].s = sc[i]; o[i].idx = i; }
    qsort(o, n, sizeof(sc_t), sc_cmp);
    const int pad = dist, gw = w + 2 * pad, gh = h + 2 * pad;
    int8_t *grid = (int8_t *)calloc((size_t)gw * gh, 1);
    for (int64_t i = 0; i < n; ++i) {
        const int x = (int)lrintf(xs[o[i].idx]), y = (int)lrintf(ys[o[i].idx]);
        grid[(size_t)(y + pad) * gw + x + pad] = 1;
    }
    int64_t cnt = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int x = (int)lrintf(xs[o[i].idx]) + pad, y = (int)lrintf(ys[o[i].idx]) + pad;
        if (grid[(size_t)y * gw + x] == 1) {
            for (int yy = y - pad; yy <= y + pad; ++yy)
                for (int xx = x - pad; xx <= x + pad; ++xx) grid[(size_t)yy * gw + xx] = 0;
            grid[(size_t)y * gw + x] = -1;
            keep[cnt++] = o[i].idx;
        }
    }
    free(o); free(grid);
    return cnt;
}
