"""Deterministic synthetic ResSegNetV2 weights and inputs.

The pretrained checkpoint named by the reference
(weights/20220810_ressegnetv2_wapv2_ce_sd2mfsf_uspg.pth, extract_localization.py:25-36)
is not shipped with the reference snapshot, so every golden vector, parity test
and benchmark in this repository runs on a seeded synthetic state_dict that has
exactly the reference's key names and shapes (nets/sfd2.py:259-303).

Only numpy RandomState + element-wise IEEE operations are used, so the bytes are
identical on every machine (the goldens in tests/golden were produced with them).

Conditioning choices (a deep random ReLU net otherwise collapses to a constant
descriptor): every conv filter with Cin >= 16 is built from +/- pairs along Cin
so it sums to exactly zero over the input channels; BN statistics are drawn around the
analytic post-conv moments; the three heads are scaled so the detector soft-max
is peaky, the three stability classes all occur and descriptors vary per pixel.
"""
import numpy as np

# name, cout, cin_per_group, k, has_bias, bn ('none' | 'plain' = affine False | 'affine')
_LAYERS = [
    ("conv1a.0", 64, 3, 3, True, ("conv1a.1", "plain")),
    ("conv1b.0", 64, 64, 3, True, ("bn1b.0", "plain")),
    ("conv2a.0", 128, 64, 3, True, ("conv2a.1", "plain")),
    ("conv2b.0", 128, 128, 3, True, ("bn2b.0", "plain")),
    ("conv3a.0", 256, 128, 3, True, ("conv3a.1", "plain")),
    ("conv3b.0", 256, 256, 3, True, ("bn3b.0", "plain")),
]
for _b in range(3):
    _LAYERS += [
        (f"conv4.{_b}.conv1", 256, 256, 1, False, (f"conv4.{_b}.bn1", "affine")),
        (f"conv4.{_b}.conv2", 256, 8, 3, False, (f"conv4.{_b}.bn2", "affine")),
        (f"conv4.{_b}.conv3", 256, 256, 1, False, (f"conv4.{_b}.bn3", "affine")),
    ]
_LAYERS += [
    ("convPa.0", 256, 256, 3, True, ("convPa.1", "affine")),
    ("convPa.3", 256, 256, 3, True, None),
    ("convDa.0", 256, 256, 3, True, ("convDa.1", "affine")),
    ("convDa.3", 256, 256, 3, True, None),
    ("convPb", 65, 256, 1, True, None),
    ("convDb", 128, 256, 1, True, None),
    ("ConvSta", 3, 256, 1, True, None),
]

_HEAD_GAIN = {"convPb": 3.0, "convDb": 1.0, "ConvSta": 1.5, "convPa.3": 1.0, "convDa.3": 1.0}


def _filter_bank(rs, cout, cin, k, gain):
    """He-scaled filters; for cin >= 8 the second half of Cin is the negated,
    rolled first half so that sum over Cin is exactly 0 for every (co, ky, kx)."""
    fan_in = cin * k * k
    std = gain * np.sqrt(2.0 / fan_in)
    if cin >= 8:
        half = rs.standard_normal((cout, cin // 2, k, k))
        w = np.concatenate([half, -np.roll(half, 1, axis=1)], axis=1)
    else:
        w = rs.standard_normal((cout, cin, k, k))
    return (w * std).astype(np.float32)


def make_state_dict(seed=0, family=None, gain_log2=0, gain_on="all"):
    """Returns {name: np.ndarray} with the reference ResSegNetV2 state_dict layout.

    family=None is the well-conditioned draw every golden vector was made with (bytes fixed).  The other families are
    for the conditioning sweep of the reduced-precision modes (tests/test_gpu_f16c_conditioning.py): weights a TRAINED
    checkpoint may have and the benign draw does not --
      "student"     heavy-tailed filters (Student-t, 3 degrees of freedom, unit variance), zero-sum as the default
      "calibrated"  the default draw with every BatchNorm's running statistics replaced by the statistics its input
                    actually has on a calibration image (what training leaves behind)
      "biased"      filters that do NOT sum to zero over the input channels (mean 0.3 sigma) + calibrated statistics
      "dead"        calibrated, and 5 % of the output channels of every BatchNorm'd layer have filters 2^-5 .. 2^-9 of
                    the others (running_var 2^-10 .. 2^-18 of theirs: the folded scale is x 32 .. x 512)
      "smallvar"    calibrated, except that on 5 % of the channels of the backbone's BatchNorms running_var is 1e-3 of the
                    input's variance (those channels come out x 31.6; the layers behind them are calibrated on that)
    gain_log2 = k multiplies the activation tensors named by gain_on ("all" backbone tensors, or one of "conv1a",
    "conv1b", "conv2a", "conv2b", "conv3a", "trunk", "t1", "t2") by 2^k and their consumers' filters by 2^-k: the
    network's function is unchanged (powers of two: exactly, up to the BatchNorm eps), only the scale of what is stored
    between the layers moves.  This is what probes the fixed scalings of the compensated mode."""
    rs = np.random.RandomState(seed)
    fam = family or "gauss"
    if fam not in ("gauss", "student", "calibrated", "biased", "dead", "smallvar"):
        raise ValueError(f"unknown weight family {family!r}")
    sd = {}
    for name, cout, cin, k, has_bias, bn in _LAYERS:
        gain = _HEAD_GAIN.get(name, 1.0)
        if fam == "student":
            sd[name + ".weight"] = _filter_bank_t(rs, cout, cin, k, gain)
        elif fam == "biased":
            w = _filter_bank(rs, cout, cin, k, gain)
            sd[name + ".weight"] = (w + 0.3 * w.std()).astype(np.float32) if cin >= 8 and name not in _HEAD_GAIN else w
        else:
            sd[name + ".weight"] = _filter_bank(rs, cout, cin, k, gain)
        if has_bias:
            sd[name + ".bias"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
        if bn is not None:
            bname, kind = bn
            sd[bname + ".running_mean"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
            sd[bname + ".running_var"] = (0.6 + 0.8 * rs.random_sample(cout)).astype(np.float32)
            sd[bname + ".num_batches_tracked"] = np.array(1, dtype=np.int64)
            if kind == "affine":
                sd[bname + ".weight"] = (1.0 + 0.1 * rs.standard_normal(cout)).astype(np.float32)
                sd[bname + ".bias"] = (0.1 * rs.standard_normal(cout)).astype(np.float32)
    if fam == "dead":
        for name, cout, cin, k, has_bias, bn in _LAYERS:
            if bn is None:
                continue
            idx = rs.choice(cout, max(1, cout // 20), replace=False)
            sh = rs.randint(5, 10, size=idx.size)
            sd[name + ".weight"][idx] *= np.exp2(-sh).astype(np.float32)[:, None, None, None]
            if has_bias:
                sd[name + ".bias"][idx] *= np.exp2(-sh).astype(np.float32)
    fixed_var = None
    if fam == "smallvar":
        # (layers that a calibrated BatchNorm follows: the stem, conv2a .. conv3b, and conv1 / conv2 of the ResBlocks -- not the skip path's
        #  bn3 nor the head branches, where x 30 would compound into the logits)
        fixed_var = {bn[0]: rs.choice(cout, max(1, cout // 20), replace=False) for name, cout, cin, k, has_bias, bn in _LAYERS
                     if bn is not None and not bn[0].endswith("bn3") and not name.startswith("convP") and not name.startswith("convD")}
    if fam in ("calibrated", "biased", "dead", "smallvar"):
        _calibrate_bn(sd, make_image(64, 96, 900 + seed), fixed_var)
    if gain_log2:
        _apply_gain(sd, int(gain_log2), gain_on)
    return sd


def _filter_bank_t(rs, cout, cin, k, gain):
    """_filter_bank with Student-t (3 degrees of freedom) entries scaled to unit variance: a few entries per filter
    are 5-20 sigma."""
    fan_in = cin * k * k
    std = gain * np.sqrt(2.0 / fan_in) / np.sqrt(3.0)
    if cin >= 8:
        half = rs.standard_t(3, size=(cout, cin // 2, k, k))
        w = np.concatenate([half, -np.roll(half, 1, axis=1)], axis=1)
    else:
        w = rs.standard_t(3, size=(cout, cin, k, k))
    return (w * std).astype(np.float32)


def _np_conv(x, w, stride, groups):
    """x [C,H,W], w [Cout,Cin/groups,k,k], zero padding k // 2 (float64 einsum over an im2col view; small maps only)."""
    cout, cg, k, _ = w.shape
    c, h, wd = x.shape
    p = k // 2
    xp = np.pad(x, ((0, 0), (p, p), (p, p)))
    ho, wo = (h + 2 * p - k) // stride + 1, (wd + 2 * p - k) // stride + 1
    cols = np.empty((c, k, k, ho, wo), dtype=np.float64)
    for dy in range(k):
        for dx in range(k):
            cols[:, dy, dx] = xp[:, dy:dy + stride * ho:stride, dx:dx + stride * wo:stride]
    og = cout // groups
    out = np.empty((cout, ho, wo), dtype=np.float64)
    for g in range(groups):
        out[g * og:(g + 1) * og] = np.einsum("oikl,iklhw->ohw", w[g * og:(g + 1) * og].astype(np.float64), cols[g * cg:(g + 1) * cg])
    return out


def _calibrate_bn(sd, img, fixed_var=None):
    """Replaces every BatchNorm's running_mean / running_var by the mean / variance its input has on `img` [3,h,w] in [0,1]
    (propagated layer by layer through the network with the statistics already replaced), floor 1e-8 on the variance.
    fixed_var: {bn name: channel indices} whose running_var is 1e-3 of the measured variance instead."""
    mean = np.array([0.485, 0.456, 0.406]).reshape(3, 1, 1)
    std = np.array([0.229, 0.224, 0.225]).reshape(3, 1, 1)

    def layer(x, conv, bn, stride=1, groups=1, relu=True, res=None):
        y = _np_conv(x, sd[conv + ".weight"], stride, groups)
        if conv + ".bias" in sd:
            y = y + sd[conv + ".bias"].astype(np.float64).reshape(-1, 1, 1)
        if bn is not None:
            m, v = y.mean(axis=(1, 2)), np.maximum(y.var(axis=(1, 2)), 1e-8)
            if fixed_var is not None and bn in fixed_var:
                v[fixed_var[bn]] *= 1e-3
            sd[bn + ".running_mean"] = m.astype(np.float32)
            sd[bn + ".running_var"] = v.astype(np.float32)
            y = (y - m.reshape(-1, 1, 1)) / np.sqrt(v.reshape(-1, 1, 1) + 1e-5)
            if bn + ".weight" in sd:
                y = y * sd[bn + ".weight"].astype(np.float64).reshape(-1, 1, 1) + sd[bn + ".bias"].astype(np.float64).reshape(-1, 1, 1)
        if res is not None:
            y = y + res
        return np.maximum(y, 0.0) if relu else y

    o = layer((img.astype(np.float64) - mean) / std, "conv1a.0", "conv1a.1")
    o = layer(o, "conv1b.0", "bn1b.0", stride=2)
    o = layer(o, "conv2a.0", "conv2a.1")
    o = layer(o, "conv2b.0", "bn2b.0", stride=2)
    o = layer(o, "conv3a.0", "conv3a.1")
    o = layer(o, "conv3b.0", "bn3b.0")
    for b in range(3):
        q = f"conv4.{b}."
        t = layer(o, q + "conv1", q + "bn1")
        t = layer(t, q + "conv2", q + "bn2", groups=32)
        o = layer(t, q + "conv3", q + "bn3", res=o)
    layer(o, "convPa.0", "convPa.1", stride=2)
    layer(o, "convDa.0", "convDa.1")


# producers / consumers of the backbone's stored tensors (gain family)
def _apply_gain(sd, k, on):
    g, gi = np.float32(2.0 ** k), np.float32(2.0 ** -k)

    def plain(conv, bn):       # BatchNorm(affine=False): y = (conv(x) + b - mean) / sqrt(var + eps)  ->  g y
        sd[conv + ".weight"] = sd[conv + ".weight"] * g
        sd[conv + ".bias"] = sd[conv + ".bias"] * g
        sd[bn + ".running_mean"] = sd[bn + ".running_mean"] * g      # (running_var unchanged: the numerator carries the factor)

    def affine(bn):            # gamma, beta
        sd[bn + ".weight"] = sd[bn + ".weight"] * g
        sd[bn + ".bias"] = sd[bn + ".bias"] * g

    def consumer(conv):        # every input channel / g
        sd[conv + ".weight"] = sd[conv + ".weight"] * gi

    chain = [("conv1a", "conv1a.0", "conv1a.1", ["conv1b.0"]), ("conv1b", "conv1b.0", "bn1b.0", ["conv2a.0"]),
             ("conv2a", "conv2a.0", "conv2a.1", ["conv2b.0"]), ("conv2b", "conv2b.0", "bn2b.0", ["conv3a.0"]),
             ("conv3a", "conv3a.0", "conv3a.1", ["conv3b.0"])]
    for name, conv, bn, cons in chain:
        if on in ("all", name):
            plain(conv, bn)
            for c in cons:
                consumer(c)
    if on in ("all", "trunk"):   # conv3b's output and every ResBlock's output share the skip path: one scale
        plain("conv3b.0", "bn3b.0")
        for b in range(3):
            affine(f"conv4.{b}.bn3")
            consumer(f"conv4.{b}.conv1")
        for c in ("convPa.0", "convDa.0", "ConvSta"):
            if c + ".weight" in sd:
                consumer(c)
    for b in range(3):
        if on in ("all", "t1"):
            affine(f"conv4.{b}.bn1")
            consumer(f"conv4.{b}.conv2")
        if on in ("all", "t2"):
            affine(f"conv4.{b}.bn2")
            consumer(f"conv4.{b}.conv3")


def make_image(h, w, seed=0):
    """[3,h,w] float32 in [0,1]: smooth low-frequency field + fine texture, so the
    detector sees structure at several scales (pure white noise gives a flat map)."""
    rs = np.random.RandomState(1000 + seed)
    img = rs.random_sample((3, h, w))
    # low-frequency component: nearest-upsampled coarse grid (exact, no float reductions)
    ch, cw = (h + 15) // 16, (w + 15) // 16
    coarse = rs.random_sample((3, ch, cw))
    low = np.repeat(np.repeat(coarse, 16, axis=1), 16, axis=2)[:, :h, :w]
    out = 0.5 * img + 0.5 * low
    return out.astype(np.float32)


def make_descriptors(n, dim=128, seed=0):
    """[n, dim] float32 unit-norm Gaussian descriptors."""
    rs = np.random.RandomState(2000 + seed)
    d = rs.standard_normal((n, dim))
    d = d / np.sqrt((d * d).sum(axis=1, keepdims=True))
    return d.astype(np.float32)
