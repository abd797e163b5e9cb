"""GPU parity tests of the compensated-fp16 mode (precision='f16c', SFD2_PREC_F16C): the throughput mode that has to
meet BASELINE.json north_star's tolerance -- descriptors within 1e-3 of the fp32 reference -- at every BASELINE geometry.

Asserted here (measured values are printed and appended to gpurun_out/f16c_parity_measured.txt when that directory exists):
  * every backbone activation (hi + residual plane) within 1e-3 * max|layer| of the fp32 oracle, the head branches' fp16
    tensors within 2e-3  (plain fp16: 1.5e-2)
  * dense and sampled descriptors within 1e-3 absolute on unit vectors (north_star), norms 1 +- 1e-5
  * detector score within 2e-2 relative (plain fp16: 8e-2)
  * key-point set IoU >= 0.985 against the fp32 oracle / the reference goldens (selection stages themselves are
    bit-exact given the heat map: tests/test_gpu_parity.py), and since round 6 the ORDER of the list (Spearman >= 0.999, bounded rank shift: _compare)

Two option sets (round 6).  `model_c` is the mode AS SHIPPED: on these weights the load-time self-check turns "c3b_plain" on (conv3b without its correction
chunks: include/sfd2_hip.h, sfd2_get_relax_status) -- every end-to-end assertion (descriptors <= 1e-3, IoU, order, goldens, every BASELINE geometry) runs on
it, and its activations behind conv3b are held to 2e-3 of max instead of 1e-3.  Tests of the compensated ARITHMETIC itself (per-activation tolerances, tuned
against generic kernels, the rb_inner / fp6 / trunk_r1 / s2d option tables with their own tighter numbers) pin "c3b_plain" = 0: `model_c_full` and every model
built inside a test.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402
from sfd2_amd import synth  # noqa: E402

DESC_TOL = 1e-3      # north_star
ACT_TOL = 1e-3       # relative to max|layer|
SCORE_TOL = 2e-2


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _record(line):
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "f16c_parity_measured.txt"), "a") as f:
            f.write(line + "\n")


@pytest.fixture(scope="module")
def model_c(synth_sd):
    if not _gpu_ok():
        pytest.fail("no MI355X visible: GPU tests cannot run (there is no CPU fallback)")
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    return m


@pytest.fixture(scope="module")
def model_c_full(synth_sd):
    """The fully compensated backbone: option c3b_plain = 0 (module docstring)."""
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.context.set_option("c3b_plain", 0)
    m.load_state_dict(synth_sd)
    m.cuda(0)
    return m


def _kp_index(kp):
    out = {}
    for i, (x, y) in enumerate(kp):
        out.setdefault((float(x), float(y)), i)
    return out


def _compare(got, want, min_iou):
    a, b = _kp_index(got["keypoints"]), _kp_index(want["keypoints"])
    common = sorted(set(a) & set(b))
    iou = len(common) / max(1, len(set(a) | set(b)))
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    dd = np.abs(got["descriptors"][ia] - np.asarray(want["descriptors"], dtype=np.float64)[ib]).max()
    gs, ws = got["scores"][ia], np.asarray(want["scores"])[ib]
    bad = np.abs(gs - ws) > SCORE_TOL * ws + 1e-4       # a stability-class flip at a near-tie changes a score by a class ratio
    shift = int(np.abs(ia - ib).max())
    same_rank = int((ia == ib).sum())
    assert dd <= DESC_TOL, dd
    assert iou >= min_iou, iou
    assert bad.mean() <= 0.005, bad.mean()
    # ORDER of the list (asserted, not only recorded).  A stability-class flip multiplies one score by a class ratio and moves that point by
    # thousands of ranks -- those points are the `bad` ones above, bounded at 0.5 %.  Among the others the two lists must agree as ordered lists: ranks taken
    # among the common, un-flipped points; Spearman correlation >= 0.999 and no point further than max(8, n / 32) ranks from its place (measured at 4096 key
    # points: <= 47; the score error of ~1e-2 relative against a score spacing of ~2e-4 relative per rank).
    good = ~bad
    if good.sum() >= 8:
        ra = np.argsort(np.argsort(ia[good])).astype(np.float64)
        rb = np.argsort(np.argsort(ib[good])).astype(np.float64)
        n = ra.size
        rho = 1.0 - 6.0 * float(((ra - rb) ** 2).sum()) / (n * (n * n - 1.0))
        worst = int(np.abs(ra - rb).max())
        assert rho >= 0.999, rho
        assert worst <= max(8, n // 32), (worst, n)
    return iou, dd, shift, same_rank, len(common)


@pytest.mark.parametrize("shipped", [False, True])
@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (37, 53, 13)])
def test_f16c_det_vs_oracle(model_c, model_c_full, synth_sd, h, w, seed, shipped):
    model_c = model_c if shipped else model_c_full
    plain3b = bool(model_c.context.get_option("c3b_plain"))
    assert plain3b == shipped, model_c.context.margin_status()       # (on these weights the self-check finds the room: 6.0-6.4e-4 on the probe against 7e-4)
    img = synth.make_image(h, w, seed)
    x = orc.norm_rgb(img)
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    score, stab, desc = model_c.det(x[None])
    worst = ("", 0.0)
    for name, want in taps.items():
        if name.endswith(".bn2"):
            continue      # the grouped conv's output stays in LDS (rb23_c_kernel); covered with fuse_rb23 = 0 below
        got = model_c.context.debug_activation(name)
        assert got.shape == want.shape, name
        err = np.abs(got - want).max() / np.abs(want).max()
        if err > worst[1]:
            worst = (name, float(err))
        # (the two head branches are plain fp16 layers whose outputs are stored as fp16 -- half an ulp alone is up to 4.9e-4 of
        #  max: 2e-3 there, plain fp16 asserts 1.5e-2)
        behind_3b = plain3b and (name == "bn3b" or name.startswith(("conv4", "ConvSta")))      # conv3b in plain fp16: its own rounding rides on the trunk
        assert err <= (2 * ACT_TOL if (behind_3b or name.startswith(("convP", "convD"))) else ACT_TOL), (name, err)
    rel = np.abs(score[0, 0] - o_score) / (o_score + 1e-4 / SCORE_TOL)
    assert rel.max() <= SCORE_TOL, rel.max()
    dd = np.abs(desc[0] - o_desc).max()
    assert dd <= DESC_TOL, dd
    np.testing.assert_allclose(np.linalg.norm(desc[0], axis=0), 1.0, atol=1e-5)
    assert (stab[0, 0] != o_stab).mean() < 0.002
    _record(f"f16c det {h}x{w} ({'as shipped: c3b_plain' if shipped else 'fully compensated'}): worst activation {worst[0]} {worst[1]:.2e} of max, score rel {rel.max():.2e}, dense desc {dd:.2e}, "
            f"stability flips {(stab[0, 0] != o_stab).mean():.2e}")


@pytest.mark.parametrize("tag", ["64x96", "100x130"])
def test_f16c_det_vs_reference_golden(model_c, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"det_{tag}.npz"), allow_pickle=False)
    img = synth.make_image(int(g["h"]), int(g["w"]), int(g["seed"]))
    score, stab, desc = model_c.det(orc.norm_rgb(img)[None])
    assert (np.abs(score[0, 0] - g["score"]) <= SCORE_TOL * g["score"] + 1e-4).all()
    dd = np.abs(desc[0] - g["desc"]).max()
    assert dd <= DESC_TOL, dd
    assert (stab[0, 0] != g["stability"]).mean() < 0.002
    _record(f"f16c det vs reference golden {tag}: dense desc {dd:.2e}")


@pytest.mark.parametrize("tag", ["96x128_k200", "100x130_all", "480x640_k1024"])
def test_f16c_extract_vs_reference_golden(model_c, golden_dir, tag):
    from sfd2_amd.extractor import extract_resnet_return
    g = np.load(os.path.join(golden_dir, f"extract_{tag}.npz"), allow_pickle=False)
    img = synth.make_image(int(g["h"]), int(g["w"]), int(g["seed"]))
    got = extract_resnet_return(model_c, img[None], conf_th=0.001, topK=int(g["topk"]), scales=[1.0])
    want = {"keypoints": g["keypoints"], "scores": g["scores"], "descriptors": g["descriptors"]}
    iou, dd, shift, same, n = _compare(got, want, 0.985)
    _record(f"f16c extract vs reference golden {tag}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


@pytest.mark.parametrize("h,w,seed,topk", [(96, 128, 21, 200), (1200, 1600, 31, 4096), (1024, 1024, 61, 4096), (768, 1024, 62, 4096),
                                            (1536, 2048, 63, 4096), (1600, 1200, 64, 4096), (1063, 1600, 65, 4096)])
def test_f16c_extract_vs_oracle(model_c, synth_sd, h, w, seed, topk):
    """Every BASELINE geometry (1600x1200 landscape and portrait, 1024x1024, 1024x768), 2048x1536, and the Aachen database
    images' 1600x1063 (height a multiple of neither 8 nor 4: resized score map, strided conv2b -- the path every database image
    of configs[2] takes); device-resident input."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(model_c, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    assert len(got["scores"]) == len(want["scores"])
    np.testing.assert_allclose(np.linalg.norm(got["descriptors"], axis=1), 1.0, atol=1e-5)
    iou, dd, shift, same, n = _compare(got, want, 0.985)
    _record(f"f16c extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


def test_f16c_reload_weights_and_mode_switch(synth_sd):
    """A second load_state_dict on a live context, and switching precision on it, must take effect in every mode
    (stale f16x3 filter splits, stale graphs)."""
    from sfd2_amd.model import ResSegNetV2
    sd2 = synth.make_state_dict(7)
    img = synth.make_image(64, 96, 11)
    x = orc.norm_rgb(img)
    for prec, tol in (("f16c", DESC_TOL), ("f16x3", 2e-5)):
        m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.det(x[None])
        m.load_state_dict(sd2)
        _, _, desc = m.det(x[None])
        _, _, o_desc = orc.det(sd2, x, {})
        assert np.abs(desc[0] - o_desc).max() <= tol, prec


@pytest.mark.parametrize("fp6", [0, 1])
@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (200, 264, 14), (37, 53, 13)])
def test_f16c_tuned_kernels_vs_generic_kernel(synth_sd, h, w, seed, fp6):
    """The tuned kernels' compensated forms (fused stem, conv3x3_pp, conv_igemm2; option 'fuse_det' routes sfd2_det through
    the fused stem) against the generic compensated kernel (option 'generic_c'), the reference implementation of the
    arithmetic: same operands, same products, fp32 summation order differs -- every backbone activation within 7e-5 of
    max|layer|.  fp6 = 1 (the default, option 'fp6_acts'): three of the tuned path's tensors carry their residuals as block-scaled
    fp6 instead of fp8 -- another rounding of a second-order term, not the same operands any more: 1.2e-4."""
    from sfd2_amd.model import ResSegNetV2
    x = orc.norm_rgb(synth.make_image(h, w, seed))
    names = ["bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "conv4.0.bn1", "conv4.0.bn2", "conv4.0", "conv4.1", "conv4.2"]
    outs = []
    for generic in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        m.context.set_option("generic_c", generic)
        m.context.set_option("rb_inner", 0)      # (the generic kernel only has the fully compensated ResBlock)
        m.context.set_option("fuse_det", 0 if generic else 1)
        m.context.set_option("fp6_acts", fp6)
        score, stab, desc = m.det(x[None])
        outs.append(({n: m.context.debug_activation(n) for n in names}, desc))
    worst = 0.0
    for n in names:
        a, b = outs[0][0][n], outs[1][0][n]
        err = np.abs(a - b).max() / np.abs(a).max()
        worst = max(worst, err)
        # (one step of a compensated tensor's corr byte is 2^-15 = 3e-5 of its binade: a flipped step after the last ResBlock reads 5.6e-5 of
        #  max once the activation exponents place the maximum at 16-32 instead of 4-8)
        assert err <= (1.2e-4 if fp6 else 7e-5), (n, err)
    dd = np.abs(outs[0][1] - outs[1][1]).max()
    assert dd <= 5e-4, dd      # (the heads are plain fp16: a one-ulp flip of a backbone hi value is fp16-level noise behind them)
    _record(f"f16c tuned vs generic {h}x{w} fp6_acts={fp6}: worst activation diff {worst:.2e} of max, dense desc diff {dd:.2e}")


@pytest.fixture(scope="module")
def model_c_rb16(synth_sd):
    """f16c with option comp_rb = 0: stem .. conv3b compensated, the three ResBlocks on the fused fp16 kernel."""
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
    m.context.set_option("comp_rb", 0)
    return m


@pytest.mark.parametrize("h,w,seed,topk", [(480, 640, 0, 1024), (1200, 1600, 31, 4096), (1024, 1024, 61, 4096), (768, 1024, 62, 4096),
                                            (1536, 2048, 63, 4096)])
def test_f16c_fp16_resblocks_extract_vs_oracle(model_c_rb16, synth_sd, h, w, seed, topk):
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(model_c_rb16, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    iou, dd, shift, same, n = _compare(got, want, 0.97)
    _record(f"f16c comp_rb=0 extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


def test_f16c_uint8_hwc_ingest_equals_float_path(model_c):
    """extract_localization.py:157-186 in the compensated mode: the uint8 HWC image (RGB, BGR, device-resident) converted
    inside the compensated fused stem equals the host-side astype(float32) / 255, bit for bit, also through the pyramid."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    rs = np.random.RandomState(7)
    base = (synth.make_image(120, 168, 9).transpose(1, 2, 0) * 255.0 + rs.uniform(-0.5, 0.5, (120, 168, 3)))
    u8 = np.clip(np.rint(base), 0, 255).astype(np.uint8)
    f = (u8.astype(np.float32).transpose(2, 0, 1) / 255.).astype(np.float32)
    want = extract_resnet_return(model_c, f[None], conf_th=0.001, topK=300, scales=[1.0])
    for arr, kw in [(u8, {}), (np.ascontiguousarray(u8[:, :, ::-1]), {"bgr": True}), (torch.from_numpy(u8).cuda(), {})]:
        got = extract_resnet_return(model_c, arr, conf_th=0.001, topK=300, scales=[1.0], **kw)
        for k in ("keypoints", "scores", "descriptors"):
            np.testing.assert_array_equal(got[k], want[k])
    want = extract_resnet_return(model_c, f[None], conf_th=0.001, topK=300, scales=[1.0, 0.75])
    got = extract_resnet_return(model_c, u8, conf_th=0.001, topK=300, scales=[1.0, 0.75])
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(got[k], want[k])


@pytest.mark.parametrize("tag,h,w,seed,topk,scales", [("96x128_k150", 96, 128, 21, 150, [1.0, 0.5])])
def test_f16c_multiscale_vs_oracle_and_reference_golden(model_c, synth_sd, golden_dir, tag, h, w, seed, topk, scales):
    """The scale pyramid of extract_resnet_return (nets/extractor.py:113-124, 211-236, 322-330) in the compensated mode."""
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    got = extract_resnet_return(model_c, img[None], conf_th=0.001, topK=topk, scales=scales)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk, scales=tuple(scales))
    iou, dd, shift, same, n = _compare(got, want, 0.97)
    g = np.load(os.path.join(golden_dir, f"extract_ms_{tag}.npz"), allow_pickle=False)
    ref = {"keypoints": g["keypoints"], "scores": g["scores"], "descriptors": g["descriptors"].astype(np.float64)}
    iou2, dd2, _, _, _ = _compare(got, ref, 0.97)
    _record(f"f16c pyramid {tag}: vs oracle IoU {iou:.4f} desc {dd:.2e}; vs reference golden IoU {iou2:.4f} desc {dd2:.2e}")


def test_f16c_spp_variant_vs_oracle(model_c, synth_sd):
    """extract.py's variant (greedy nms_fast, extract_spp_feats_singlescale: extract.py:17-84, 204-277) on the compensated
    conv stack: same points up to near-threshold ones, descriptors within 1e-3."""
    from sfd2_amd.extract import extract_spp_feats_singlescale
    x = orc.norm_rgb(synth.make_image(96, 128, 21))
    pts, desc, sc = extract_spp_feats_singlescale(model_c, x[None], conf_th=0.1)[:3]
    wpts, wdesc, wsc = orc.extract_spp_feats_singlescale(synth_sd, x, conf_th=0.1)[:3]
    a = {(float(p[0]), float(p[1])): i for i, p in enumerate(np.asarray(pts)[:, :2])}
    b = {(float(p[0]), float(p[1])): i for i, p in enumerate(np.asarray(wpts)[:, :2])}
    common = sorted(set(a) & set(b))
    assert len(common) >= 0.97 * max(len(a), len(b))
    dd = max(np.abs(np.asarray(desc)[a[k]] - np.asarray(wdesc)[b[k]]).max() for k in common)
    assert dd <= DESC_TOL, dd
    _record(f"f16c extract.py variant 128x96: {len(common)} of {len(b)} points in common, desc {dd:.2e}")


def test_f16c_hipgraph_cache_equals_eager(model_c):
    """sfd2_extract_match with option 'graphs' in the compensated mode (what bench.py's headline replays): captured and
    replayed units equal the eager calls bit for bit, and a precision switch in between drops the cached graphs."""
    import ctypes
    import torch
    from sfd2_amd import _lib
    ctx = model_c.context
    lib = ctx.lib
    H, W, K, KDB, N = 480, 640, 2048, 4, 2048
    imgs = [torch.from_numpy(synth.make_image(H, W, 71 + i)).cuda() for i in range(2)]
    db = [torch.from_numpy(synth.make_descriptors(N, seed=80 + i)).to(torch.float16).cuda().contiguous() for i in range(KDB)]
    dbs = (_lib.DescSet * KDB)(*[_lib.DescSet(d.data_ptr(), N, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in db])
    mconf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)
    kp = torch.zeros((K, 2), device="cuda"); sc = torch.zeros((K,), device="cuda"); de = torch.zeros((K, 128), device="cuda")
    mt = torch.full((KDB, K), -7, dtype=torch.int64, device="cuda"); ms = torch.zeros((KDB, K), device="cuda")

    def unit(i):
        _lib.check(lib.sfd2_extract_match(ctx.h, imgs[i].data_ptr(), H, W, 0.001, K, 0, kp.data_ptr(), sc.data_ptr(), de.data_ptr(),
                                          dbs, KDB, 128, ctypes.byref(mconf), mt.data_ptr(), ms.data_ptr()))
        ctx.sync()
        return tuple(t.clone() for t in (kp, sc, de, mt, ms))

    want = [unit(0), unit(1)]
    ctx.set_option("graphs", 1)
    try:
        for rnd in range(3):
            for i in (0, 1):
                got = unit(i)
                assert all(torch.equal(g, w) for g, w in zip(got, want[i])), (rnd, i)
        ctx.set_precision("f16")          # (a cached graph must not survive a precision switch)
        f16 = unit(0)
        assert not torch.equal(f16[2], want[0][2])
        ctx.set_precision("f16c")
        got = unit(0)
        assert all(torch.equal(g, w) for g, w in zip(got, want[0]))
        # option "sta_side" (round 6): ConvSta as a fork inside the captured graph -- the same bits, eager and replayed, images alternating
        ctx.set_option("sta_side", 1)
        for rnd in range(3):
            for i in (0, 1):
                got = unit(i)
                assert all(torch.equal(g, w) for g, w in zip(got, want[i])), ("sta_side", rnd, i)
        ctx.set_option("graphs", 0)
        for i in (0, 1):
            got = unit(i)
            assert all(torch.equal(g, w) for g, w in zip(got, want[i])), ("sta_side eager", i)
    finally:
        ctx.set_option("graphs", 0)
        ctx.set_option("sta_side", 0)
        ctx.set_precision("f16c")


@pytest.fixture(scope="module")
def model_c_heads(synth_sd):
    """f16c at its most precise: option comp_heads = 1 (the 3x3 layers of the two head branches compensated as well) and
    rb_inner = 0 (every tensor of the ResBlocks compensated)."""
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
    m.context.set_option("comp_heads", 1)
    m.context.set_option("rb_inner", 0)
    return m


@pytest.mark.parametrize("h,w,seed,topk", [(100, 130, 22, -1), (480, 640, 0, 1024), (1200, 1600, 31, 4096)])
def test_f16c_compensated_heads_extract_vs_oracle(model_c_heads, synth_sd, h, w, seed, topk):
    """With the head branches compensated too the conv stack's only fp16-class roundings left are convPb / convDb:
    descriptors within 5e-4 (CPU-twin prediction 1.3e-4), key-point sets equal up to a handful of near-ties."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(model_c_heads, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    iou, dd, shift, same, n = _compare(got, want, 0.99)
    assert dd <= 5e-4, dd
    _record(f"f16c comp_heads=1 extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


def test_f16c_compensated_heads_det_vs_oracle(model_c_heads, synth_sd):
    x = orc.norm_rgb(synth.make_image(100, 130, 12))
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    score, stab, desc = model_c_heads.det(x[None])
    for name in ("convPa", "convDa"):
        got = model_c_heads.context.debug_activation(name)
        err = np.abs(got - taps[name]).max() / np.abs(taps[name]).max()
        assert err <= 6e-4, (name, err)      # (the branch outputs are stored as plain fp16: half an ulp is up to 4.9e-4 of max)
    dd = np.abs(desc[0] - o_desc).max()
    assert dd <= 5e-4, dd
    _record(f"f16c comp_heads=1 det 130x100: dense desc {dd:.2e}")


@pytest.mark.parametrize("h,w,seed,topk", [(100, 130, 22, -1), (480, 640, 0, 1024), (1200, 1600, 31, 4096)])
def test_f16c_fp6_filters_extract_vs_oracle(synth_sd, h, w, seed, topk):
    """Option fp6_filters = 1: the correction filters of conv2a / conv3a / conv3b as e2m3 with one power-of-two scale per output
    channel (fp8 x fp6 scaled MFMA; operand layout pinned by tools/probe/mfma_fp6_layout.hip).  Same tolerance as the default."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
    m.context.set_option("fp6_filters", 1)
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(m, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    iou, dd, shift, same, n = _compare(got, want, 0.985)
    assert dd <= 7e-4, dd
    _record(f"f16c fp6_filters=1 extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


@pytest.fixture(scope="module")
def model_c_fp6(synth_sd):
    if not _gpu_ok():
        pytest.fail("no MI355X visible: GPU tests cannot run (there is no CPU fallback)")
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
    m.context.set_option("fp6_acts", 1)
    return m


@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (37, 53, 13)])
def test_f16c_fp6_acts_det_vs_oracle(model_c_fp6, synth_sd, h, w, seed):
    """Option fp6_acts = 1: the corr records of conv1b's, conv2b's and conv3a's output as block-scaled fp6 half-records
    (v_cvt_scalef32_2xpk16_fp6_f32 in the producers' epilogues, fp6 x fp6 scaled MFMA in conv2a / conv3a / conv3b).  Every stored
    tensor -- the three fp6 ones decoded by sfd2_debug_activation from their half-records -- within the default's 1e-3 of max."""
    img = synth.make_image(h, w, seed)
    x = orc.norm_rgb(img)
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    score, stab, desc = model_c_fp6.det(x[None])
    worst = ("", 0.0)
    seen6 = 0
    for name, want in taps.items():
        if name.endswith(".bn2"):
            continue
        got = model_c_fp6.context.debug_activation(name)
        err = np.abs(got - want).max() / np.abs(want).max()
        if name in ("bn1b", "bn2b", "conv3a"):
            seen6 += 1
            # the residual these records carry is worth keeping: the hi plane alone is off by half an fp16 ulp (up to 4.9e-4 of max)
            assert err <= 2.5e-4, (name, err)
        if err > worst[1]:
            worst = (name, float(err))
        assert err <= (2 * ACT_TOL if name.startswith(("convP", "convD")) else ACT_TOL), (name, err)
    assert seen6 == 3
    dd = np.abs(desc[0] - o_desc).max()
    assert dd <= DESC_TOL, dd
    rel = np.abs(score[0, 0] - o_score) / (o_score + 1e-4 / SCORE_TOL)
    assert rel.max() <= SCORE_TOL, rel.max()
    assert (stab[0, 0] != o_stab).mean() < 0.002
    st = model_c_fp6.range_status()
    assert not st["saturated"], st
    _record(f"f16c fp6_acts=1 det {h}x{w}: worst activation {worst[0]} {worst[1]:.2e} of max, score rel {rel.max():.2e}, dense desc {dd:.2e}")


@pytest.mark.parametrize("h,w,seed,topk", [(100, 130, 22, -1), (96, 128, 21, 200), (480, 640, 0, 1024), (1200, 1600, 31, 4096), (1024, 1024, 61, 4096),
                                           (333, 517, 5, 300)])
def test_f16c_fp6_acts_extract_vs_oracle(model_c_fp6, synth_sd, h, w, seed, topk):
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(model_c_fp6, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    iou, dd, shift, same, n = _compare(got, want, 0.985)
    _record(f"f16c fp6_acts=1 extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


def test_f16c_fp6_acts_per_tensor_fallbacks(synth_sd):
    """The format is decided per tensor: without the fused stem conv1b's output stays fp8 (conv2a reads fp8 records, conv3a / conv3b fp6),
    without conv3x3_rf<2,comp> conv2b's does; generic_c switches the whole option off.  All of them inside the tolerance, and the
    combination with fp6_filters / compensated heads too."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    img = synth.make_image(100, 130, 22)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=-1)
    for opts in ({"fuse": 0}, {"no_rf_c": 1}, {"generic_c": 1}, {"fp6_filters": 1}, {"comp_heads": 1}, {"rb_inner": 0}, {"comp_rb": 0}, {"branches": 1}):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        m.context.set_option("fp6_acts", 1)
        for k, v in opts.items():
            m.context.set_option(k, v)
        got = extract_resnet_return(m, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=-1, scales=[1.0])
        iou, dd, shift, same, n = _compare(got, want, 0.985)
        _record(f"f16c fp6_acts=1 {opts} extract 130x100: IoU {iou:.4f}, desc {dd:.2e}")


@pytest.mark.parametrize("h,w,seed,topk", [(96, 128, 21, 200), (480, 640, 0, 1024), (1200, 1600, 31, 4096), (1600, 1200, 64, 4096), (1024, 1024, 61, 4096),
                                           (100, 132, 22, -1), (36, 40, 3, -1), (1028, 772, 7, 2000)])
def test_f16c_conv2b_space_to_depth_vs_oracle_and_strided_kernel(synth_sd, h, w, seed, topk):
    """Option s2d (default on; throughput path, image sides multiples of 4): conv2a stores its output as four parity planes at quarter
    resolution and conv2b runs as a stride-1 layer over them (conv2b_s2d_kernel.hip) -- the same products as conv3x3_rf<2,comp>, another
    fp32 summation order: both inside the tolerance against the oracle, and within 5e-4 of each other on the common key points (the bound of
    test_f16c_tuned_kernels_vs_generic_kernel: a flipped fp16 rounding of a backbone value is fp16-level noise behind the plain fp16 heads)."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    outs = []
    for s2d in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        m.context.set_option("s2d", s2d)
        got = extract_resnet_return(m, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
        iou, dd, shift, same, n = _compare(got, want, 0.985 if topk > 0 else 0.97)      # (every point above the threshold: a few sit on it)
        outs.append(got)
        st = m.range_status()
        assert not st["saturated"] and st["fallbacks"] == 0, st
        _record(f"f16c s2d={s2d} extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}")
    a, b = _kp_index(outs[0]["keypoints"]), _kp_index(outs[1]["keypoints"])
    common = sorted(set(a) & set(b))
    assert len(common) >= 0.98 * min(len(a), len(b))
    d = np.abs(outs[0]["descriptors"][[a[k] for k in common]] - outs[1]["descriptors"][[b[k] for k in common]]).max()
    assert d <= 5e-4, d
    _record(f"f16c s2d on vs off {w}x{h}: {len(common)} common key points, descriptors within {d:.2e}")


@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (200, 264, 14)])
def test_f16c_trunk_residual_only_tensors(synth_sd, h, w, seed):
    """Option trunk_r1 (default on): conv3b's output and the ResBlocks' outputs carry one correction byte per channel (the residual) instead
    of the (residual, value) unit; conv1x1_c256_c rebuilds the value byte from the hi plane.  Every tensor of the trunk -- the three-byte
    ones decoded by sfd2_debug_activation -- within the mode's tolerance of the oracle.  Against the four-byte form: conv3b's output identical
    (same hi plane, same residual bytes); behind it the rebuilt value byte is e4m3(hi / 4) where the stored one was e4m3(y / 4), which moves a
    sum by ~1e-6 and flips a few fp16 roundings of the plain tensor t1 -- one fp16 ulp at the top binade is 4.9e-4 of max."""
    from sfd2_amd.model import ResSegNetV2
    x = orc.norm_rgb(synth.make_image(h, w, seed))
    taps = {}
    orc.det(synth_sd, x, taps)
    names = ["bn3b", "conv4.0.bn1", "conv4.0", "conv4.1", "conv4.2"]      # (the oracle taps the first block's inner tensors)
    outs = []
    for r1 in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        m.context.set_option("trunk_r1", r1)
        m.det(x[None])
        got = {n: m.context.debug_activation(n) for n in names}
        for n in names:
            err = np.abs(got[n] - taps[n]).max() / np.abs(taps[n]).max()
            assert err <= ACT_TOL, (r1, n, err)
        outs.append(got)
    worst = max(np.abs(outs[0][n] - outs[1][n]).max() / np.abs(outs[1][n]).max() for n in names)
    assert np.array_equal(outs[0]["bn3b"], outs[1]["bn3b"])
    assert worst <= 6e-4, worst
    _record(f"f16c trunk_r1 on vs off {h}x{w}: trunk activations within {worst:.2e} of max")


@pytest.fixture(scope="module", params=[0, 1])
def model_c_inner(request, synth_sd):
    """f16c with option rb_inner off its default (2: the tensors inside the ResBlocks, t1 and t2, stored as plain fp16):
    1 = only t2 plain, 0 = both compensated."""
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
    m.context.set_option("rb_inner", request.param)
    m.rb_inner = request.param
    return m


@pytest.mark.parametrize("h,w,seed,topk", [(100, 130, 22, -1), (480, 640, 0, 1024), (1200, 1600, 31, 4096), (1024, 1024, 61, 4096),
                                            (768, 1024, 62, 4096), (1536, 2048, 63, 4096), (1600, 1200, 64, 4096)])
def test_f16c_rb_inner_extract_vs_oracle(model_c_inner, synth_sd, h, w, seed, topk):
    """CPU-twin prediction (dense / sampled): rb_inner = 0 3.0e-4 / 3.2e-4, 1 3.8e-4 / 3.2e-4, 2 (the default, covered by every
    other test of this file) 5.2e-4 / 4.1e-4.  Asserted: north_star's 1e-3 as everywhere, and 5e-4 for these two settings."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    got = extract_resnet_return(model_c_inner, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=topk, scales=[1.0])
    iou, dd, shift, same, n = _compare(got, want, 0.985)
    assert dd <= 5e-4, dd
    _record(f"f16c rb_inner={model_c_inner.rb_inner} extract {w}x{h} top{topk}: IoU {iou:.4f}, desc {dd:.2e}, same rank {same}/{n}, max rank shift {shift}")


def test_f16c_rb_inner_det_vs_oracle(model_c_inner, synth_sd):
    x = orc.norm_rgb(synth.make_image(100, 130, 12))
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    score, stab, desc = model_c_inner.det(x[None])
    for b in range(3):
        got = model_c_inner.context.debug_activation(f"conv4.{b}")
        err = np.abs(got - taps[f"conv4.{b}"]).max() / np.abs(taps[f"conv4.{b}"]).max()
        assert err <= ACT_TOL, (b, err)
    dd = np.abs(desc[0] - o_desc).max()
    assert dd <= DESC_TOL, dd
    _record(f"f16c rb_inner={model_c_inner.rb_inner} det 130x100: dense desc {dd:.2e}")


@pytest.mark.parametrize("h,w,topk", [(100, 130, -1), (480, 640, 1024), (1200, 1600, 4096), (333, 517, 300), (1030, 770, 2000)])
def test_f16c_fused_conv2_conv3_bit_identical(synth_sd, h, w, topk):
    """rb23_c_kernel (ResBlock.conv2 + conv3 + residual in one launch, t2 in LDS; option 'fuse_rb23', default on with
    rb_inner = 2) performs the operations of the two kernels it replaces in the same order: identical bits, on the arena path
    of sfd2_extract (where the block's output must not take t1's slot any more), for tiles cut by the right and bottom edges
    and for several tiles per block."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    img = torch.from_numpy(synth.make_image(h, w, 78)).cuda()
    x = orc.norm_rgb(synth.make_image(min(h, 200), min(w, 264), 14))
    outs = []
    for fuse in (0, 1):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        m.context.set_option("fuse_rb23", fuse)
        m.context.set_option("trunk_r1", 0)      # (the same tensor format on both sides: the residual-only trunk tensors need the fused kernel)
        o = extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0])
        m.det(x[None])
        o["acts"] = [m.context.debug_activation(f"conv4.{b}") for b in range(3)]
        if fuse == 0:      # two launches: the grouped conv's output is in HBM and can be compared with the oracle's
            taps = {}
            orc.det(synth_sd, x, taps)
            for nm in ("conv4.0.bn1", "conv4.0.bn2"):      # (the oracle taps the first block's inner tensors)
                got = m.context.debug_activation(nm)
                err = np.abs(got - taps[nm]).max() / np.abs(taps[nm]).max()
                assert err <= ACT_TOL, (nm, err)
        else:
            with pytest.raises(RuntimeError, match="not materialised"):
                m.context.debug_activation("conv4.0.bn2")
        outs.append(o)
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(outs[0][k], outs[1][k])
    for a, b in zip(outs[0]["acts"], outs[1]["acts"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("prec,tol", [("f16c", DESC_TOL), ("f16", 3e-3)])
@pytest.mark.parametrize("h,w,seed,topk", [(100, 130, 22, 60), (480, 640, 0, 1024), (1200, 1600, 31, 4096), (333, 517, 5, 300), (1600, 1200, 64, 4096)])
def test_sparse_da3_equals_dense_path_and_oracle(synth_sd, prec, tol, h, w, seed, topk):
    """Option 'sparse_da3' (default on, extract path): convDa.3 is computed on the 4 x K bilinear corner pixels of the selected key
    points only (sparse_da3_kernel), instead of on the whole 1/4-resolution map.  Same products as the dense layer, another fp32
    summation order: key points and scores are untouched (they do not depend on the descriptor branch), descriptors agree with
    the dense path to 1e-4 (an fp16 store of convDa.3's output now and then rounds the other way) and with the oracle within the
    mode's tolerance.  Corners outside the map (key points on the border) and key-point counts that do not fill a block are in
    the small cases."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    img = synth.make_image(h, w, seed)
    dev = torch.from_numpy(img)[None].cuda()
    outs = []
    for sp in (0, 1):
        m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("sparse_da3", sp)
        outs.append(extract_resnet_return(m, dev, conf_th=0.001, topK=topk, scales=[1.0]))
    np.testing.assert_array_equal(outs[0]["keypoints"], outs[1]["keypoints"])
    np.testing.assert_array_equal(outs[0]["scores"], outs[1]["scores"])
    dd = np.abs(outs[0]["descriptors"] - outs[1]["descriptors"]).max()
    assert dd <= 1e-4, dd
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    a, b = _kp_index(outs[1]["keypoints"]), _kp_index(want["keypoints"])
    common = sorted(set(a) & set(b))
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    do = np.abs(outs[1]["descriptors"][ia] - np.asarray(want["descriptors"], dtype=np.float64)[ib]).max()
    assert do <= tol, do
    np.testing.assert_allclose(np.linalg.norm(outs[1]["descriptors"], axis=1), 1.0, atol=1e-5)
    _record(f"{prec} sparse_da3 {w}x{h} top{topk}: vs dense path {dd:.2e}, vs oracle {do:.2e} ({len(common)} common key points)")


def test_f16c_detector_branch_compensation_table(synth_sd):
    """Which head layers to compensate: the detector score goes through exp(), so the detector branch's fp16
    roundings are what moves key points.  Records score error / key-point agreement for {default, comp_det, comp_heads}; asserts
    that comp_det buys what it is for (a smaller score error than the default, key points no worse)."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    x = orc.norm_rgb(synth.make_image(100, 130, 12))
    o_score, _, _ = orc.det(synth_sd, x, {})
    img = synth.make_image(1200, 1600, 31)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=4096)
    rows = {}
    for name, opts in (("default", {}), ("comp_det", {"comp_det": 1}), ("comp_heads", {"comp_heads": 1})):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("c3b_plain", 0)      # the arithmetic under test is the fully compensated backbone (module docstring)
        for k, v in opts.items():
            m.context.set_option(k, v)
        score, _, _ = m.det(x[None])
        rel = float((np.abs(score[0, 0] - o_score) / (o_score + 1e-4 / SCORE_TOL)).max())
        got = extract_resnet_return(m, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=4096, scales=[1.0])
        iou, dd, shift, same, n = _compare(got, want, 0.985)
        rows[name] = (rel, iou, same, n, dd)
        _record(f"f16c head compensation [{name}]: score rel {rel:.2e}, 1600x1200 IoU {iou:.4f}, same rank {same}/{n}, max rank shift {shift}, desc {dd:.2e}")
    # Measured (round 4; CHANGELOG) and predicted by the CPU twin (profiles/r04_detector_budget.txt): the score error
    # (1.2-1.3e-2 relative) is the BACKBONE's f16c-level error carried through the detector branch into logits of magnitude ~18 and
    # through exp() -- compensating convPa.0 / convPa.3, even with convPb in three passes, moves it by a quarter at most.  The option is
    # therefore off by default; what is asserted is that it does no harm.
    assert rows["comp_det"][0] <= 1.5 * rows["default"][0] and rows["comp_heads"][0] <= 1.5 * rows["default"][0], rows
    assert rows["comp_det"][1] >= rows["default"][1] - 2e-3, rows
