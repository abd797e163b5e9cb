/*
 * ORACLE (test infrastructure, NOT product code) -- CPU restatement of the dense
 * arithmetic on the SFD2 hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library.
 *
 * This file: the convolution / batch-norm primitives the reference obtains from
 * torch.nn.Conv2d / BatchNorm2d (nets/sfd2.py:14-22, :58-95, :259-303).
 * Compiled with FP contraction allowed (speed); everything whose result feeds an
 * exact comparison lives in orc_post.c (contraction off).
 *
 * Parity pinning: checked against tests/golden/det_*.npz, which were produced by
 * importing the reference nets/sfd2.py in the authoring container
 * (tests/golden/gen_goldens.py).  The reference repository itself ships no tests
 * or golden vectors for this path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[co][oy][ox] = bias[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] * in[g*cpg+ci][oy*s+ky-p][ox*s+kx-p]
 * torch.nn.Conv2d semantics: zero padding p = k/2 (every conv on the path uses
 * padding = k//2, nets/sfd2.py:14-22,68-95,286-303), groups as in conv3x3(groups=32).
 * Accumulation order per output element: bias, then (ci, ky, kx) ascending, fp32. */
void orc_conv2d(const float *in, int cin, int h, int w,
                const float *wt, const float *bias, int cout, int k, int stride, int groups,
                float *out)
{
    const int pad = k / 2;
    const int ho = (h + 2 * pad - k) / stride + 1;
    const int wo = (w + 2 * pad - k) / stride + 1;
    const int cpg = cin / groups;   /* input channels per group */
    const int opg = cout / groups;  /* output channels per group */
#pragma omp parallel for collapse(2) schedule(static)
    for (int co = 0; co < cout; ++co) {
        for (int oy = 0; oy < ho; ++oy) {
            float *orow = out + ((size_t)co * ho + oy) * wo;
            const float b = bias ? bias[co] : 0.0f;
            for (int ox = 0; ox < wo; ++ox) orow[ox] = b;
            const int g = co / opg;
            for (int ci = 0; ci < cpg; ++ci) {
                const float *ip = in + (size_t)(g * cpg + ci) * h * w;
                const float *wp = wt + ((size_t)co * cpg + ci) * k * k;
                for (int ky = 0; ky < k; ++ky) {
                    const int iy = oy * stride + ky - pad;
                    if (iy < 0 || iy >= h) continue;
                    const float *irow = ip + (size_t)iy * w;
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wp[ky * k + kx];
                        /* ox range with 0 <= ox*stride + kx - pad < w */
                        int ox0 = 0;
                        while (ox0 * stride + kx - pad < 0) ++ox0;
                        int ox1 = wo;
                        while (ox1 > ox0 && (ox1 - 1) * stride + kx - pad >= w) --ox1;
                        const float *ib = irow + kx - pad;
                        if (stride == 1) {
                            for (int ox = ox0; ox < ox1; ++ox) orow[ox] += wv * ib[ox];
                        } else {
                            for (int ox = ox0; ox < ox1; ++ox) orow[ox] += wv * ib[ox * stride];
                        }
                    }
                }
            }
        }
    }
}

/* BatchNorm2d inference (running stats, eps) + optional affine + optional residual
 * add + optional ReLU, in place on x[c][hw].  nets/sfd2.py:58-65 (affine=False),
 * :25-55 (ResBlock: bn(affine) [+ identity] + relu).
 * y = (x - mean) / sqrt(var + eps) * gamma + beta  */
void orc_bn_act(float *x, int c, size_t hw, const float *mean, const float *var,
                const float *gamma, const float *beta, float eps,
                const float *residual, int relu)
{
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch) {
        const float inv = 1.0f / sqrtf(var[ch] + eps);
        const float a = gamma ? inv * gamma[ch] : inv;
        const float b = (beta ? beta[ch] : 0.0f) - mean[ch] * a;
        float *p = x + (size_t)ch * hw;
        const float *r = residual ? residual + (size_t)ch * hw : NULL;
        for (size_t i = 0; i < hw; ++i) {
            float v = p[i] * a + b;
            if (r) v += r[i];
            if (relu && v < 0.0f) v = 0.0f;
            p[i] = v;
        }
    }
}

void orc_relu(float *x, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (x[i] < 0.0f) x[i] = 0.0f;
}

/* sim[n][m] = sum_d a[n][d] * b[m][d]   (fp32; hloc/matchers/nearest_neighbor.py:39-40) */
void orc_sim_f32(const float *a, int n, const float *b, int m, int d, float *sim)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            float s = 0.0f;
            for (int t = 0; t < d; ++t) s += a[(size_t)i * d + t] * b[(size_t)j * d + t];
            sim[(size_t)i * m + j] = s;
        }
}

/* fp64 flavour (it_loc/matcher.py:124: descriptors1 @ descriptors2.t() on float64) */
void orc_sim_f64(const double *a, int n, const double *b, int m, int d, double *sim)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int t = 0; t < d; ++t) s += a[(size_t)i * d + t] * b[(size_t)j * d + t];
            sim[(size_t)i * m + j] = s;
        }
}
