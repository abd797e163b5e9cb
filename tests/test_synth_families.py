"""The synthetic weight families of the conditioning sweep (sfd2_amd/synth.py), checked on the CPU oracle: the default draw keeps the
bytes every golden vector was made with, the gain family leaves the network's function alone (that is what makes it a probe of the
number formats and of nothing else), and every family gives a network that still detects key points."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as orc
from sfd2_amd import synth


def _digest(sd):
    return hashlib.sha256(b"".join(np.ascontiguousarray(sd[k]).tobytes() for k in sorted(sd))).hexdigest()


def test_default_draw_is_unchanged():
    assert _digest(synth.make_state_dict(0)) == "a6e8fc35ab2fd8629ab7d0ef03a4e5ef76141eb5219bcb388e70b2cc66836ef1"
    assert _digest(synth.make_state_dict(0, family=None, gain_log2=0)) == _digest(synth.make_state_dict(0))


@pytest.mark.parametrize("k,on", [(6, "all"), (-6, "all"), (10, "trunk"), (-8, "conv2a"), (8, "t1"), (-8, "t2")])
def test_gain_family_preserves_the_function(k, on):
    """powers of two through conv / BatchNorm / ReLU / the skip path: same descriptors, same scores (fp32 rounding only)"""
    img = synth.make_image(48, 64, 3)
    x = orc.norm_rgb(img)
    s0, st0, d0 = orc.det(synth.make_state_dict(0), x, None)
    taps = {}
    s1, st1, d1 = orc.det(synth.make_state_dict(0, gain_log2=k, gain_on=on), x, taps)
    assert np.abs(d1 - d0).max() < 2e-5
    assert np.abs(s1 - s0).max() < 2e-5 * max(1.0, float(np.abs(s0).max()))
    assert (st1 != st0).mean() < 1e-3
    if on == "trunk":
        t0 = {}
        orc.det(synth.make_state_dict(0), x, t0)
        np.testing.assert_allclose(taps["conv4.2"], t0["conv4.2"] * 2.0 ** k, rtol=1e-4, atol=1e-4 * 2.0 ** k)


@pytest.mark.parametrize("family", ["student", "calibrated", "biased", "dead", "smallvar"])
def test_families_give_working_networks(family):
    sd = synth.make_state_dict(1, family=family)
    assert set(sd) == set(synth.make_state_dict(0))
    img = synth.make_image(96, 128, 7)
    out = orc.extract_resnet_return(sd, img, conf_th=0.001, topK=200)
    assert len(out["keypoints"]) >= 100, (family, len(out["keypoints"]))
    norms = np.linalg.norm(out["descriptors"], axis=1)
    np.testing.assert_allclose(norms, 1.0, atol=1e-5)
    if family == "dead":
        w = sd["conv2a.0.weight"]
        mx = np.abs(w).reshape(w.shape[0], -1).max(axis=1)
        assert mx.min() < mx.max() / 16          # some channels' filters really are 2^-5 .. 2^-9 of the others
    if family == "smallvar":
        t = {}
        orc.det(sd, orc.norm_rgb(img), t)
        assert np.abs(t["conv2a"]).max() > 60     # x 31.6 channels
