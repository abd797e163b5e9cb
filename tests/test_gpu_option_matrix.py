"""Walks the legal option x precision matrix of sfd2_set_option at 96x128 against the fp32 oracle ("19 options x
4 precisions ... is where the next silent wrong-kernel bug will come from").  Every option is flipped on its own in every precision it
applies to, plus the pairs that route through combined code (compensated heads x fp6 filters, rb_inner x fuse_rb23, branches x the
sparse descriptor head ...).  One extraction per row, the tolerance of the row's precision:
    f32 / f16x3   descriptors <= 2e-5, key-point set IoU >= 0.99
    f16c          descriptors <= 1e-3 (north_star), IoU >= 0.97
    f16           descriptors <= 3e-3, IoU >= 0.90
Options that must not change a bit (alias, graphs-independent launch forms, cu_limit, fuse_post / fuse_pb / sparse_desc) are ALSO checked
for bit-identity against the same precision's default row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402
from sfd2_amd import synth  # noqa: E402

TOL = {"f32": (2e-5, 0.99), "f16x3": (2e-5, 0.99), "f16c": (1e-3, 0.97), "f16": (3e-3, 0.90)}
ALL = ("f32", "f16x3", "f16c", "f16")
F16S = ("f16c", "f16")

# (options, precisions, bit-identical to the precision's default row?)
ROWS = [
    ({}, ALL, False),
    ({"fuse": 0}, F16S, False),
    ({"alias": 0}, ("f16",), True),
    ({"alias": 0}, ("f16c",), False),    # (private buffers are the readable layout: conv2b on the strided kernel, not the space-to-depth one)
    ({"fuse_post": 0}, ALL, True),
    ({"fuse_pb": 0}, F16S, False),
    ({"sparse_desc": 0}, F16S, False),
    ({"sparse_da3": 0}, ("f16c", "f16", "f16x3"), False),
    ({"sparse_desc": 0, "sparse_da3": 0}, ("f16x3",), False),
    ({"branches": 1}, F16S, False),      # (no sparse convDa.3 beside a side stream: another fp32 summation order)
    ({"branches": 1, "sparse_desc": 0}, F16S, False),
    ({"cu_limit": 64}, ALL, True),
    ({"sta_side": 1}, F16S, True),           # (round 6: ConvSta on the side stream: the same kernel, the same bits)
    ({"sta_side": 1, "branches": 1}, F16S, False),      # ("branches" owns the side stream: sta_side steps back)
    ({"cu_limit": 7, "fuse": 0}, F16S, False),
    ({"x3_pp": 0}, ("f16x3",), False),
    ({"auto_range": 0}, F16S, False),
    ({"comp_rb": 0}, ("f16c",), False),
    ({"rb_inner": 0}, ("f16c",), False),
    ({"rb_inner": 1}, ("f16c",), False),
    ({"rb_inner": 1, "fuse_rb23": 0}, ("f16c",), False),
    ({"fuse_rb23": 0}, ("f16c",), False),     # (bit-identical to the fused kernel at the same tensor format: test_f16c_fused_conv2_conv3_bit_identical)
    ({"comp_heads": 1}, ("f16c",), False),
    ({"comp_det": 1}, ("f16c",), False),
    ({"comp_heads": 1, "fp6_filters": 1}, ("f16c",), False),
    ({"comp_heads": 1, "comp_rb": 0}, ("f16c",), False),
    ({"comp_det": 1, "branches": 1}, ("f16c",), False),
    ({"fp6_filters": 1}, ("f16c",), False),
    ({"generic_c": 1}, ("f16c",), False),
    ({"generic_c": 1, "rb_inner": 0, "fuse": 0}, ("f16c",), False),
    ({"no_rf_c": 1}, ("f16c",), False),
    ({"range_fallback": 0}, ("f16c",), True),
    ({"fp6_acts": 1}, ("f16c",), False),
    ({"fp6_acts": 1, "fuse": 0}, ("f16c",), False),
    ({"fp6_acts": 1, "no_rf_c": 1}, ("f16c",), False),
    ({"fp6_acts": 1, "generic_c": 1}, ("f16c",), False),
    ({"fp6_acts": 1, "fp6_filters": 1, "comp_heads": 1}, ("f16c",), False),
    ({"trunk_r1": 0}, ("f16c",), False),
    ({"trunk_r1": 1, "comp_heads": 1}, ("f16c",), False),     # (the last block's output keeps its units for the compensated heads)
    ({"trunk_r1": 1, "fp6_acts": 0}, ("f16c",), False),       # (falls back: conv3b's residual-only store exists for its fp6-input form)
    ({"trunk_r1": 1, "fuse_rb23": 0}, ("f16c",), False),
    ({"s2d": 0}, ("f16c",), False),
    ({"s2d": 1, "fp6_acts": 0}, ("f16c",), False),       # (the s2d store exists for the fp6-input form of conv2a only: falls back)
    ({"s2d": 1, "no_rf_c": 1}, ("f16c",), False),
]


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="module")
def case(synth_sd):
    if not _gpu_ok():
        pytest.fail("no MI355X visible: GPU tests cannot run (there is no CPU fallback)")
    img = synth.make_image(96, 128, 21)
    return img, orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=200)


_default = {}


def _run(sd, img, prec, opts):
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
    m.cuda(0)
    pre = {k: v for k, v in opts.items() if k == "auto_range"}           # (takes effect at load time)
    for k, v in pre.items():
        m.context.set_option(k, v)
    m.load_state_dict(sd)
    for k, v in opts.items():
        if k not in pre:
            m.context.set_option(k, v)
    return extract_resnet_return(m, img[None], conf_th=0.001, topK=200, scales=[1.0]), m


@pytest.mark.parametrize("prec", ALL)
def test_option_matrix(synth_sd, case, prec):
    img, want = case
    tol, min_iou = TOL[prec]
    wk = {(float(x), float(y)): i for i, (x, y) in enumerate(want["keypoints"])}
    base = None
    n_rows = 0
    for opts, precs, identical in ROWS:
        if prec not in precs:
            continue
        got, m = _run(synth_sd, img, prec, opts)
        gk = {(float(x), float(y)): i for i, (x, y) in enumerate(got["keypoints"])}
        common = sorted(set(gk) & set(wk))
        iou = len(common) / max(1, len(set(gk) | set(wk)))
        dd = max(np.abs(got["descriptors"][gk[k]] - want["descriptors"][wk[k]]).max() for k in common)
        assert iou >= min_iou, (prec, opts, iou)
        assert dd <= tol, (prec, opts, dd)
        if prec == "f16c":
            st = m.range_status()
            assert not st["saturated"] and st["fallbacks"] == 0, (opts, st)
        if not opts:
            base = got
        elif identical:
            for key in ("keypoints", "scores", "descriptors"):
                np.testing.assert_array_equal(got[key], base[key], err_msg=f"{prec} {opts} {key}")
        n_rows += 1
    assert n_rows >= 3


def test_unknown_option_and_bad_values_are_errors():
    from sfd2_amd import _lib
    ctx = _lib.Context(0)
    with pytest.raises(RuntimeError):
        ctx.set_option("no_such_option", 1)
    ctx.set_option("rb_inner", 7)            # clamped to 2, not an error (documented)
    ctx.set_option("cu_limit", -3)           # 0 = off
    ctx.close()
