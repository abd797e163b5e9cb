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


def make_state_dict(seed=0):
    """Returns {name: np.ndarray} with the reference ResSegNetV2 state_dict layout."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, cout, cin, k, has_bias, bn in _LAYERS:
        gain = _HEAD_GAIN.get(name, 1.0)
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
    return sd


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
