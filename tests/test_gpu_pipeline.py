"""GPU tests of the pipelined host drivers (sfd2_amd/pipeline.py): the decode-pool / asynchronous-extract / writer loop of
extract_localization.main and the query-grouped, device-resident loop of match_features.main write exactly what the
reference-shaped serial loops write (dataset by dataset, bit for bit); plus the C-ABI additions they stand on
(sfd2_extract_record_async, sfd2_desc_pack, SFD2_FLAG_ASYNC with host outputs) and the round-4 ADVICE items on the
range bookkeeping."""
import ctypes
import os

import numpy as np
import pytest

from oracle import oracle as orc
from sfd2_amd import _lib, synth

pytestmark = pytest.mark.gpu


def _gpu_ok():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _model(sd, precision, **opts):
    if not _gpu_ok():
        pytest.fail("no MI355X visible: GPU tests cannot run (there is no CPU fallback)")
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision=precision).eval()
    m.cuda(0)
    for k, v in opts.items():
        m.context.set_option(k, v)
    m.load_state_dict(sd)
    return m


def _stores_equal(a_path, b_path):
    from sfd2_amd.feature_io import open_store
    a, b = open_store(a_path, "r"), open_store(b_path, "r")
    assert list(a.keys()) == list(b.keys()) and len(list(a.keys())) > 0, (list(a.keys()), list(b.keys()))
    for k in a.keys():
        assert sorted(a[k].keys()) == sorted(b[k].keys()), k
        for ds in b[k].keys():
            x, y = np.asarray(a[k][ds].__array__()), np.asarray(b[k][ds].__array__())
            assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), (k, ds)
    return list(a.keys())


@pytest.mark.parametrize("precision", ["f16c", "f16x3"])
def test_pipelined_extract_equals_serial_loop(tmp_path, synth_sd, precision):
    """files -> feature store: num_workers = 3 (decoder threads, asynchronous extracts, writer threads) against the
    serial loop, on images of four geometries, two of which go through the device cubic resize (resize_max 160)."""
    pytest.importorskip("PIL")
    from PIL import Image
    from sfd2_amd import extract_localization as el
    root = tmp_path / "images"
    (root / "db").mkdir(parents=True)
    (root / "query").mkdir(parents=True)
    sizes = [(120, 160), (160, 120), (150, 210), (96, 128), (201, 140), (120, 160), (128, 160), (160, 128), (120, 160), (96, 128), (144, 160)]
    for i, (h, w) in enumerate(sizes):
        u8 = (synth.make_image(h, w, 300 + i).transpose(1, 2, 0) * 255).astype(np.uint8)
        Image.fromarray(u8).save(root / ("db" if i % 3 else "query") / f"im{i:02d}.png")
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 300}, "preprocessing": {"grayscale": False, "resize_max": 160}}
    model = _model(synth_sd, precision)
    ds = el.ImageDataset(root, conf["preprocessing"])
    assert len(ds) == len(sizes)
    p_serial = el.main(conf, ds, tmp_path / "serial", model_and_extractor=(model, el.extract_resnet_return), num_workers=0)
    p_pipe = el.main(conf, ds, tmp_path / "pipe", model_and_extractor=(model, el.extract_resnet_return), num_workers=3, writers=2, depth=3)
    names = _stores_equal(p_pipe, p_serial)
    assert len(names) == len(sizes)
    # two lanes: a replica context (own stream) takes every other image -- same store
    p_two = el.main(conf, ds, tmp_path / "two", model_and_extractor=(model, el.extract_resnet_return), num_workers=4, depth=4, lanes=2)
    _stores_equal(p_two, p_serial)
    # tag filter and the two-rank sharding go through the same loop
    p_tag_s = el.main(conf, ds, tmp_path / "tag_s", model_and_extractor=(model, el.extract_resnet_return), tag="query", num_workers=0)
    p_tag_p = el.main(conf, ds, tmp_path / "tag_p", model_and_extractor=(model, el.extract_resnet_return), tag="query", num_workers=2)
    assert all(n.startswith("query/") for n in _stores_equal(p_tag_p, p_tag_s))
    assert model.context.range_status()["fallbacks"] == 0


def test_pipelined_extract_in_memory_items_and_float_images(tmp_path, synth_sd):
    """Items that are not files: uint8 arrays (asynchronous path) mixed with float [3,H,W] images (synchronous call
    between the same decode and writer stages); item order is kept across the two kinds."""
    from sfd2_amd import extract_localization as el
    model = _model(synth_sd, "f16c")
    items = []
    for i in range(7):
        f = synth.make_image(96, 128, 400 + i)
        if i in (2, 5):
            items.append({"name": f"m/{i}.png", "image": f, "original_size": (128, 96)})
        else:
            items.append({"name": f"m/{i}.png", "image": (f.transpose(1, 2, 0) * 255).astype(np.uint8), "original_size": (256, 192)})
    # a constant image: whatever the detector makes of it (possibly no key point at all), both loops must store the same group
    items.append({"name": "m/flat.png", "image": np.full((96, 128, 3), 128, dtype=np.uint8), "original_size": (128, 96)})
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 150}}
    a = el.main(conf, items, tmp_path / "s", model_and_extractor=(model, el.extract_resnet_return), num_workers=0)
    b = el.main(conf, items, tmp_path / "p", model_and_extractor=(model, el.extract_resnet_return), num_workers=2)
    assert _stores_equal(b, a) == sorted(it["name"] for it in items)
    # no key point anywhere (threshold above every score): empty groups from both loops (the reference raises in np.vstack here)
    conf0 = {**conf, "model": {**conf["model"], "conf_th": 10.0}}
    a0 = el.main(conf0, items, tmp_path / "s0", model_and_extractor=(model, el.extract_resnet_return), num_workers=0)
    b0 = el.main(conf0, items, tmp_path / "p0", model_and_extractor=(model, el.extract_resnet_return), num_workers=2)
    _stores_equal(b0, a0)
    from sfd2_amd.feature_io import open_store
    g = open_store(b0, "r")["m/0.png"]
    assert g["keypoints"].shape == (0, 2) and g["descriptors"].shape == (128, 0) and g["scores"].shape == (0,)


def test_pipelined_extract_repeats_saturated_images_in_strict_mode(tmp_path):
    """SFD2_PREC_F16C, exponents zero, weights x 2^10: every image saturates.  The per-image record reports it, the
    pipelined loop repeats each such image synchronously (-> SFD2_PREC_F16X3 inside the library), and the store equals
    the serial loop's (whose synchronous extracts fall back the same way)."""
    from sfd2_amd import extract_localization as el
    sd = synth.make_state_dict(0, gain_log2=10)
    model = _model(sd, "f16c", auto_range=0, range_fallback=1)
    items = [{"name": f"s/{i}.png", "image": (synth.make_image(96, 128, 500 + i).transpose(1, 2, 0) * 255).astype(np.uint8),
              "original_size": (128, 96)} for i in range(4)]
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 100}}
    a = el.main(conf, items, tmp_path / "s", model_and_extractor=(model, el.extract_resnet_return), num_workers=0)
    n_serial = model.context.range_status(reset=True)["fallbacks"]
    b = el.main(conf, items, tmp_path / "p", model_and_extractor=(model, el.extract_resnet_return), num_workers=2)
    st = model.context.range_status()
    assert n_serial == 4 and st["fallbacks"] == 8, (n_serial, st)
    _stores_equal(b, a)
    want = orc.extract_resnet_return(sd, items[0]["image"].transpose(2, 0, 1).astype(np.float32) / 255.0, conf_th=0.001, topK=100)
    from sfd2_amd.feature_io import open_store
    kp = open_store(b, "r")["s/0.png"]["keypoints"].__array__()
    assert len({tuple(p) for p in kp} & {tuple(p) for p in want["keypoints"]}) >= 0.98 * len(want["keypoints"])


def test_extract_record_async_and_host_outputs(synth_sd):
    """SFD2_FLAG_ASYNC with (pinned) host outputs + sfd2_extract_record_async == the synchronous call."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    model = _model(synth_sd, "f16c")
    ctx = model.context
    K = 256
    imgs = [np.ascontiguousarray((synth.make_image(96, 128, 600 + i).transpose(1, 2, 0) * 255).astype(np.uint8)) for i in range(3)]   # (C order: the product is laid out like its CHW operand)
    want = [extract_resnet_return(model, im, conf_th=0.001, topK=K) for im in imgs]
    pins = [torch.from_numpy(im).pin_memory() for im in imgs]
    outs = []
    for p in pins:
        kp = torch.empty((K, 2), dtype=torch.float32, pin_memory=True)
        sc = torch.empty((K,), dtype=torch.float32, pin_memory=True)
        de = torch.empty((K, 128), dtype=torch.float32, pin_memory=True)
        rec = torch.zeros(4, dtype=torch.int32, pin_memory=True)
        n = ctypes.c_int(0)
        _lib.check(ctx.lib.sfd2_extract(ctx.h, p.data_ptr(), 0, 96, 128, 0.001, K, _lib.FLAG_ASYNC | _lib.FLAG_IMG_U8_HWC,
                                        kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 0, K, ctypes.byref(n)))
        assert n.value == -1
        _lib.check(ctx.lib.sfd2_extract_record_async(ctx.h, rec.data_ptr(), 0))
        outs.append((kp, sc, de, rec))
    ctx.sync()
    for (kp, sc, de, rec), w in zip(outs, want):
        n = int(rec[0])
        assert n == len(w["keypoints"]) and int(rec[1]) >= n and int(rec[2]) == 0 and int(rec[3]) == 0
        np.testing.assert_array_equal(kp.numpy()[:n].astype(np.float64), w["keypoints"])
        np.testing.assert_array_equal(sc.numpy()[:n].astype(np.float64), w["scores"])
        np.testing.assert_array_equal(de.numpy()[:n].astype(np.float64), w["descriptors"])
    # the records moved the maxima into the history: the status still shows them, nothing is saturated
    st = ctx.range_status()
    assert not st["saturated"] and max(v["max_stored"] for v in st["tensors"].values()) > 0.0


def test_recalibration_clears_recorded_maxima():
    """maxima recorded under other exponents say nothing about the new scaling -- sfd2_set_act_exponents /
    sfd2_calibrate_range clear the running words and both histories."""
    import torch
    sd = synth.make_state_dict(0)
    model = _model(sd, "f16c")
    ctx = model.context
    u8 = np.ascontiguousarray((synth.make_image(96, 128, 700).transpose(1, 2, 0) * 255).astype(np.uint8))
    e0, _ = ctx.act_exponents()
    ctx.set_act_exponents(e0 + 12)                       # everything 4096 x larger as stored: saturates
    K = 64
    kp = torch.empty((K, 2), dtype=torch.float32, device="cuda")
    sc = torch.empty((K,), dtype=torch.float32, device="cuda")
    de = torch.empty((K, 128), dtype=torch.float32, device="cuda")
    rec = torch.zeros(4, dtype=torch.int32, pin_memory=True)
    n = ctypes.c_int(0)
    p = torch.from_numpy(u8).cuda()
    for with_record in (False, True):                    # running words only / folded into the device-side history
        _lib.check(ctx.lib.sfd2_extract(ctx.h, p.data_ptr(), 1, 96, 128, 0.001, K, _lib.FLAG_ASYNC | _lib.FLAG_IMG_U8_HWC,
                                        kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
        if with_record:
            _lib.check(ctx.lib.sfd2_extract_record_async(ctx.h, rec.data_ptr(), 0))
        ctx.sync()
        assert ctx.range_status()["saturated"]
        if with_record:
            assert int(rec[2]) != 0
        ctx.set_act_exponents(e0 + 12)                   # same values: still a new scaling as far as the records go
        st = ctx.range_status()
        assert not st["saturated"] and all(v["max_stored"] == 0.0 for v in st["tensors"].values()), st
    ctx.set_act_exponents(e0)


def test_async_leftover_does_not_trigger_fallback(synth_sd):
    """The leftover case proper: two contexts' worth of state in one -- an asynchronous extract of a huge-gain image saturates,
    then a synchronous extract of a normal image must return the compensated mode's own result, not the strict repeat."""
    import torch
    from sfd2_amd.extractor import extract_resnet_return
    model = _model(synth_sd, "f16c")
    ctx = model.context
    ok = (synth.make_image(96, 128, 710).transpose(1, 2, 0) * 255).astype(np.uint8)
    ref = extract_resnet_return(model, ok, conf_th=0.001, topK=64)
    assert ctx.range_status(reset=True)["fallbacks"] == 0
    # an image far outside the calibrated range: float input x 4000 (the network is linear up to the first ReLU, BatchNorm keeps the gain)
    hot = (synth.make_image(96, 128, 711) * 4000.0).astype(np.float32)
    K = 64
    kp = torch.empty((K, 2), dtype=torch.float32, device="cuda")
    sc = torch.empty((K,), dtype=torch.float32, device="cuda")
    de = torch.empty((K, 128), dtype=torch.float32, device="cuda")
    n = ctypes.c_int(0)
    p = torch.from_numpy(hot).cuda()
    _lib.check(ctx.lib.sfd2_extract(ctx.h, p.data_ptr(), 1, 96, 128, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(),
                                    de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()
    got = extract_resnet_return(model, ok, conf_th=0.001, topK=64)
    st = ctx.range_status()
    assert st["fallbacks"] == 0, st                                  # the synchronous call was judged on its own image
    if not st["saturated"]:
        pytest.skip("the x 4000 image did not saturate on this checkpoint family: nothing to guard")
    np.testing.assert_array_equal(got["keypoints"], ref["keypoints"])
    np.testing.assert_array_equal(got["descriptors"], ref["descriptors"])


def _feature_store(path, sets):
    from sfd2_amd.feature_io import open_store, write_features
    st = open_store(path, "w")
    for name, d in sets.items():
        n = d.shape[0]
        write_features(st, name, {"keypoints": np.zeros((n, 2)), "descriptors": d.astype(np.float64).transpose(),
                                  "scores": np.zeros((n,)), "image_size": np.array([640, 480])})
    st.close()


@pytest.mark.parametrize("conf_name", ["NNM", "NNR", "ONN"])
def test_grouped_match_driver_equals_per_pair_loop(tmp_path, conf_name):
    """feature store -> match store: query-grouped, device-resident sets, batched launches (with a cache so small that
    sets are evicted and packed again) against the reference-shaped per-pair loop.  Sets of different sizes, duplicate
    and reversed pairs, a query that is also a database image."""
    from sfd2_amd import match_features as mf
    rs = np.random.RandomState(5)
    sizes = {"query/q0.jpg": 700, "query/q1.jpg": 1024, "query/q2.jpg": 333, "db/a.jpg": 1024, "db/b.jpg": 512, "db/c.jpg": 37,
             "db/d.jpg": 900, "db/e.jpg": 1024, "db/f.jpg": 4}
    base = synth.make_descriptors(1024, seed=9)
    sets = {}
    for i, (name, n) in enumerate(sizes.items()):
        d = base[rs.permutation(1024)[:n]] + 0.05 * rs.standard_normal((n, 128)).astype(np.float32)
        sets[name] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    feats = "feats-x"
    _feature_store(str(tmp_path / (feats + ".h5")), sets)
    dbs = [n for n in sizes if n.startswith("db/")]
    pairs = [f"{q} {d}" for q in sizes if q.startswith("query/") for d in dbs]
    pairs += ["db/a.jpg db/b.jpg", "db/b.jpg db/a.jpg", "db/a.jpg db/d.jpg", "query/q0.jpg db/a.jpg", "db/e.jpg query/q1.jpg"]
    rs.shuffle(pairs)
    a = mf.main(mf.confs[conf_name], pairs, feats, tmp_path, pairs_name="serial", grouped=False)
    b = mf.main(mf.confs[conf_name], pairs, feats, tmp_path, pairs_name="grouped", grouped=True, cache_bytes=6 * 1024 * 128 * 2)
    keys = _stores_equal(b, a)
    assert len(keys) == len(mf.unique_pairs(pairs))
    from sfd2_amd.feature_io import open_store
    g = open_store(b, "r")[mf.names_to_pair("query/q1.jpg", "db/a.jpg")]
    assert g["matches0"].dtype == np.int16 and g["matching_scores0"].dtype == np.float16 and (g["matches0"].__array__() >= 0).mean() > 0.5
    # a second run finds every pair stored and does nothing (hloc/match_features.py:93-94)
    b2 = mf.main(mf.confs[conf_name], pairs, feats, tmp_path, pairs_name="grouped", grouped=True)
    assert b2 == b and _stores_equal(b2, a) == keys


def test_desc_pack_bit_identical_to_inline_conversion():
    """sfd2_desc_pack is the conversion sfd2_match* runs on its inputs: packed sets (fp16 [n][128], resident) give the
    same matches and scores as the float64 / float32 originals in either layout, with and without a row selection."""
    import torch
    ctx = _lib.default_context(0)
    rs = np.random.RandomState(3)
    d0 = synth.make_descriptors(900, seed=1)
    d1 = synth.make_descriptors(777, seed=2)
    conf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)

    def run(q, db):
        m = np.empty((q.n,), dtype=np.int64)
        s = np.empty((q.n,), dtype=np.float32)
        _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(q), ctypes.byref(db), 1, 128, ctypes.byref(conf), m.ctypes.data,
                                            s.ctypes.data, 0, 0))
        return m, s

    rows = np.sort(rs.permutation(777)[:300]).astype(np.int32)
    for dtype, code in ((np.float64, _lib.DT_F64), (np.float32, _lib.DT_F32)):
        for layout in (_lib.LAYOUT_ND, _lib.LAYOUT_DN):
            a0 = np.ascontiguousarray((d0 if layout == _lib.LAYOUT_ND else d0.T).astype(dtype))
            a1 = np.ascontiguousarray((d1 if layout == _lib.LAYOUT_ND else d1.T).astype(dtype))
            for r in (None, rows):
                q = _lib.DescSet(a0.ctypes.data, 900, code, layout, 0, None, 0, 0)
                db = _lib.DescSet(a1.ctypes.data, 777, code, layout, 0, None if r is None else r.ctypes.data, 0 if r is None else len(r), 0)
                want = run(q, db)
                p0 = torch.empty((900, 128), dtype=torch.float16, device="cuda")
                p1 = torch.empty((777 if r is None else len(r), 128), dtype=torch.float16, device="cuda")
                _lib.check(ctx.lib.sfd2_desc_pack(ctx.h, ctypes.byref(q), 128, p0.data_ptr(), 0))
                _lib.check(ctx.lib.sfd2_desc_pack(ctx.h, ctypes.byref(db), 128, p1.data_ptr(), _lib.FLAG_ASYNC))
                ctx.sync()
                got = run(_lib.DescSet(p0.data_ptr(), 900, _lib.DT_F16, _lib.LAYOUT_ND, 1, None, 0, 0),
                          _lib.DescSet(p1.data_ptr(), p1.shape[0], _lib.DT_F16, _lib.LAYOUT_ND, 1, None, 0, 0))
                if r is not None:          # packed sets report positions in the packed set; the row-selected call reports the caller's rows
                    hit = got[0] >= 0
                    got[0][hit] = r[got[0][hit]]
                np.testing.assert_array_equal(got[0], want[0])
                np.testing.assert_array_equal(got[1], want[1])
    # the packed bytes are fp16(float(x)) row-major
    np.testing.assert_array_equal(p0.cpu().numpy(), d0.astype(np.float32).astype(np.float16))


def test_f16x3_debug_activation_after_throughput_extract(synth_sd):
    """VERDICT r4 weak #10 / ADVICE r3: after an f16x3 sfd2_extract on its throughput kernels (tensors exist as hi / lo' planes
    only) sfd2_debug_activation must refuse the tensors that were not written as fp32 -- not hand back what an earlier sfd2_det
    left in the buffers -- and still serve the ones that were."""
    from sfd2_amd.extractor import extract_resnet_return
    model = _model(synth_sd, "f16x3")
    ctx = model.context
    img = synth.make_image(96, 128, 800)
    x = orc.norm_rgb(img)
    model.det(x[None])                                             # parity entry point: every tensor readable
    before = ctx.debug_activation("conv3a")
    assert np.isfinite(before).all() and before.shape == (256, 24, 32)
    extract_resnet_return(model, synth.make_image(96, 128, 801)[None], conf_th=0.001, topK=100)   # another image, throughput path
    for name in ("conv1a", "bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "conv4.0", "conv4.1", "conv4.0.bn1", "conv4.2.bn2", "convDa.0", "convDa"):
        with pytest.raises(RuntimeError, match="not materialised"):
            ctx.debug_activation(name)
    taps = {}
    orc.det(synth_sd, orc.norm_rgb(synth.make_image(96, 128, 801)), taps)
    got = ctx.debug_activation("conv4.2")
    assert np.abs(got - taps["conv4.2"]).max() <= 2e-5 * max(1.0, np.abs(taps["conv4.2"]).max())
    model.det(x[None])                                             # and the parity entry point makes them readable again
    np.testing.assert_array_equal(ctx.debug_activation("conv3a"), before)


def test_f16_matcher_decides_like_fp64_above_2e4_gap_on_network_descriptors(synth_sd):
    """SURVEY 8c asks for identical matches where the top-1 / top-2 gap exceeds 1e-4; SFD2_SIM_F16's worst-case
    bound only promises 1e-3.  On descriptors the NETWORK produces (two views of a scene: true correspondences and near-duplicates)
    its measured similarity error is ~6e-5 (tools/match_gap_stats.py, profiles/r05c_match_gap_stats.json: 0 of 2 160 rows with a gap in
    (1e-4, 1e-3] decided differently) -- asserted here with a 2x margin: no arg-max differs where the fp64 gap exceeds 2e-4, and
    SFD2_SIM_F16X2 meets 1e-5."""
    from sfd2_amd.extractor import extract_resnet_return
    model = _model(synth_sd, "f16c")
    ctx = model.context
    base = synth.make_image(480, 640, 11)
    rs = np.random.RandomState(3)
    views = []
    for v in range(3):
        img = np.roll(base, (7 * v, -11 * v), axis=(1, 2))
        if v:
            img = np.clip(img * (1.0 + 0.05 * v) + 0.02 * rs.standard_normal(img.shape).astype(np.float32), 0.0, 1.0).astype(np.float32)
        views.append(extract_resnet_return(model, img[None], conf_th=0.001, topK=2048)["descriptors"].astype(np.float32))
    n_band = 0
    for v in (1, 2):
        for d0, d1 in ((views[0], views[v]), (views[v], views[0])):
            sim = d0.astype(np.float64) @ d1.astype(np.float64).T
            order = np.argsort(-sim, axis=1)[:, :2]
            rows = np.arange(len(sim))
            gap = sim[rows, order[:, 0]] - sim[rows, order[:, 1]]
            for mode, min_gap in ((_lib.SIM_F16, 2e-4), (_lib.SIM_F16X2, 1e-5)):
                conf = _lib.MatchConf(_lib.MATCH_HLOC, 0, 0.0, 0.0, mode)
                m = np.empty((len(d0),), dtype=np.int64)
                s = np.empty((len(d0),), dtype=np.float32)
                a0, a1 = np.ascontiguousarray(d0), np.ascontiguousarray(d1)
                _lib.check(ctx.lib.sfd2_match(ctx.h, a0.ctypes.data, len(a0), a1.ctypes.data, len(a1), 128, _lib.DT_F32, _lib.LAYOUT_ND, 0,
                                              ctypes.byref(conf), m.ctypes.data, s.ctypes.data, 0))
                sel = gap > min_gap
                assert (m[sel] == order[sel, 0]).all(), (mode, int((m[sel] != order[sel, 0]).sum()))
                assert np.abs(s - (sim.max(1) + 1.0) / 2.0).max() <= (1e-3 if mode == _lib.SIM_F16 else 1e-5)
            n_band += int(((gap > 2e-4) & (gap <= 1e-3)).sum())
    assert n_band > 50      # the band the contract is about is populated on these sets


@pytest.mark.parametrize("mode", ["NNM", "NNR"])
def test_store_matcher_equals_feature_matching_batch(tmp_path, mode):
    """The localiser's loop against a feature store with resident database sets (sfd2_amd.localize.StoreMatcher) == feature_matching_batch on
    the arrays read back from the store, with 3D-point masks, the <= 3 early-out, an unmasked image, a tiny cache (evictions) and a query that
    is itself a set of the store."""
    from sfd2_amd.feature_io import open_store
    from sfd2_amd.localize import StoreMatcher, feature_matching_batch
    from sfd2_amd.matcher import Matcher, confs as mconfs
    rs = np.random.RandomState(21)
    base = synth.make_descriptors(1500, seed=31)
    sets, ids = {}, {}
    for i, n in enumerate([900, 64, 1500, 10, 700, 333]):
        d = base[rs.permutation(1500)[:n]] + 0.03 * rs.standard_normal((n, 128)).astype(np.float32)
        sets[f"db/{i}.jpg"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        ids[f"db/{i}.jpg"] = np.where(rs.random_sample(n) < 0.5, rs.randint(0, 100000, n), -1)
    ids["db/3.jpg"][:] = -1
    ids["db/3.jpg"][:3] = 7
    sets["query/q.jpg"] = base[:800].copy()
    _feature_store(str(tmp_path / "feats.h5"), sets)
    feats = open_store(str(tmp_path / "feats.h5"), "r")
    mt = Matcher(mconfs[mode]).eval().cuda()
    sm = StoreMatcher(mt, feats, cache_bytes=1500 * 128 * 2)      # less than one query's sets: every call evicts
    names = [n for n in sets if n.startswith("db/")]
    q = feats["query/q.jpg"]["descriptors"].__array__().transpose()                  # [N,128] float64, as the localiser reads it
    dbs = [np.ascontiguousarray(feats[n]["descriptors"].__array__().transpose()) for n in names]
    for trial, id_list in enumerate(([ids[n] for n in names], [None if i == 2 else ids[n] for i, n in enumerate(names)], None)):
        want = feature_matching_batch(q, dbs, mt, [None] * len(names) if id_list is None else id_list)
        got = sm.match(q, names, id_list)
        got_by_name = sm.match("query/q.jpg", names, id_list)
        for i in range(len(names)):
            np.testing.assert_array_equal(got[i], want[i], err_msg=f"{trial} {names[i]}")
            np.testing.assert_array_equal(got_by_name[i], want[i], err_msg=f"{trial} {names[i]} (query by name)")
    assert (sm.match(q, names, [ids[n] for n in names])[3] == -1).all()
    assert sm.sets.evictions > 0 and sm.sets.loads > len(names)
    sm.close()


def test_grouped_match_driver_with_empty_and_tiny_sets(tmp_path):
    """Images without key points (n = 0) and with a handful, on either side of a pair: the grouped driver and the per-pair loop agree (all -1 /
    empty rows), nothing is read out of bounds."""
    from sfd2_amd import match_features as mf
    sets = {"q/empty.jpg": np.zeros((0, 128), np.float32), "q/some.jpg": synth.make_descriptors(300, seed=41),
            "db/empty.jpg": np.zeros((0, 128), np.float32), "db/one.jpg": synth.make_descriptors(1, seed=42),
            "db/many.jpg": synth.make_descriptors(513, seed=43)}
    _feature_store(str(tmp_path / "feats-e.h5"), sets)
    pairs = [f"{q} {d}" for q in ("q/empty.jpg", "q/some.jpg") for d in ("db/empty.jpg", "db/one.jpg", "db/many.jpg")] + ["db/one.jpg db/many.jpg"]
    a = mf.main(mf.confs["NNM"], pairs, "feats-e", tmp_path, pairs_name="serial", grouped=False)
    b = mf.main(mf.confs["NNM"], pairs, "feats-e", tmp_path, pairs_name="grouped", grouped=True)
    assert len(_stores_equal(b, a)) == 7
    from sfd2_amd.feature_io import open_store
    st = open_store(b, "r")
    assert st[mf.names_to_pair("q/empty.jpg", "db/many.jpg")]["matches0"].shape == (0,)
    assert (st[mf.names_to_pair("q/some.jpg", "db/empty.jpg")]["matches0"].__array__() == -1).all()
    assert st[mf.names_to_pair("q/some.jpg", "db/empty.jpg")]["matches0"].shape == (300,)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16c", "f16x3"])
@pytest.mark.parametrize("h,w,K,cap", [(96, 128, 200, 200), (250, 333, 4096, 4096), (480, 640, 1024, 1500)])
def test_desc_store64_is_the_float_descriptors_cast_and_transposed(synth_sd, precision, h, w, K, cap):
    """SFD2_FLAG_DESC_STORE64: `desc` comes back as double [128][cap_out] -- what extract_localization.py:253,269-272 store (transpose, float64) -- with exactly the
    values of the float [n][128] form (an fp32 is an fp64) and zero columns behind the count; host and device destinations, synchronous and asynchronous; an image
    with fewer key points than the capacity (250x333: ~1400 of 4096) and a capacity above top_k.  The matcher unit and the pyramid refuse the flag."""
    import torch
    m = _model(synth_sd, precision)
    ctx = m.context
    img = synth.make_image(h, w, 77)

    def run(flags, on_dev, async_):
        kp = np.zeros((cap, 2), np.float32); sc = np.zeros((cap,), np.float32)
        de = np.full((128, cap), -7.0, np.float64) if flags & _lib.FLAG_DESC_STORE64 else np.zeros((cap, 128), np.float32)
        n = ctypes.c_int(0)
        if on_dev:
            tk, ts, td = torch.from_numpy(kp).cuda(), torch.from_numpy(sc).cuda(), torch.from_numpy(de).cuda()
            ptrs = (tk.data_ptr(), ts.data_ptr(), td.data_ptr())
        else:
            ptrs = (kp.ctypes.data, sc.ctypes.data, de.ctypes.data)
        _lib.check(ctx.lib.sfd2_extract(ctx.h, img.ctypes.data, 0, h, w, 0.001, K, flags | (_lib.FLAG_ASYNC if async_ else 0), ptrs[0], ptrs[1], ptrs[2],
                                        1 if on_dev else 0, cap, ctypes.byref(n)))
        ctx.sync()
        if async_:
            _lib.check(ctx.lib.sfd2_extract_count(ctx.h, ctypes.byref(n)))
        if on_dev:
            kp, sc, de = tk.cpu().numpy(), ts.cpu().numpy(), td.cpu().numpy()
        return n.value, kp, sc, de
    n, kp, sc, de = run(0, False, False)
    assert 20 < n <= K
    for on_dev in (False, True):
        for async_ in (False, True):
            n2, kp2, sc2, d64 = run(_lib.FLAG_DESC_STORE64, on_dev, async_)
            assert n2 == n
            np.testing.assert_array_equal(kp2[:n], kp[:n]); np.testing.assert_array_equal(sc2[:n], sc[:n])
            assert d64.dtype == np.float64 and d64.shape == (128, cap)
            np.testing.assert_array_equal(d64[:, :n], de[:n].T.astype(np.float64))
            assert not d64[:, n:].any()
    sc1 = (ctypes.c_double * 1)(1.0)
    rc = ctx.lib.sfd2_extract_multiscale(ctx.h, img.ctypes.data, 0, h, w, sc1, 1, ctypes.c_float(0.001), K, _lib.FLAG_DESC_STORE64, kp.ctypes.data, sc.ctypes.data, None, 0, cap, None)
    assert rc != 0 and b"SFD2_FLAG_DESC_STORE64" in ctx.lib.sfd2_last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", [(96, 128), (250, 333), (1200, 1600)])
def test_rgbx_pixels_equal_rgb_pixels(synth_sd, h, w):
    """SFD2_FLAG_IMG_U8_X (four bytes per pixel, PIL's in-memory layout, the fourth ignored) through sfd2_extract from host and from device memory,
    and through sfd2_preprocess: the outputs of the three-byte form, bit for bit."""
    import torch
    from sfd2_amd.model import ResSegNetV2
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(synth_sd)
    m.cuda(0)
    ctx = m.context
    rgb = (synth.make_image(h, w, 31).transpose(1, 2, 0) * 255).astype(np.uint8).copy()
    rgbx = np.concatenate([rgb, np.full((h, w, 1), 0xA5, np.uint8)], axis=2).copy()
    K = 300
    def run(arr, flags, on_dev):
        kp = np.zeros((K, 2), np.float32); sc = np.zeros((K,), np.float32); de = np.zeros((K, 128), np.float32)
        n = ctypes.c_int(0)
        src = torch.from_numpy(arr).cuda() if on_dev else None
        ptr = src.data_ptr() if on_dev else arr.ctypes.data
        _lib.check(ctx.lib.sfd2_extract(ctx.h, ptr, 1 if on_dev else 0, h, w, 0.001, K, _lib.FLAG_IMG_U8_HWC | flags, kp.ctypes.data, sc.ctypes.data,
                                        de.ctypes.data, 0, K, ctypes.byref(n)))
        return n.value, kp, sc, de
    ref = run(rgb, 0, False)
    assert ref[0] > 20
    for on_dev in (False, True):
        got = run(rgbx, _lib.FLAG_IMG_U8_X, on_dev)
        assert got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            np.testing.assert_array_equal(a, b)
    # the resize path
    nh, nw = max(8, h // 2), max(8, w // 2)
    o3 = torch.empty((3, nh, nw), device="cuda"); o4 = torch.empty((3, nh, nw), device="cuda")
    _lib.check(ctx.lib.sfd2_preprocess(ctx.h, rgb.ctypes.data, 0, h, w, 0, nh, nw, o3.data_ptr()))
    _lib.check(ctx.lib.sfd2_preprocess(ctx.h, rgbx.ctypes.data, 0, h, w, _lib.FLAG_IMG_U8_X, nh, nw, o4.data_ptr()))
    assert torch.equal(o3, o4)
    # a float image cannot carry the flag, and the pyramid entry point does not take four-byte pixels
    assert ctx.lib.sfd2_extract(ctx.h, rgbx.ctypes.data, 0, h, w, 0.001, K, _lib.FLAG_IMG_U8_X, None, None, None, 0, K, None) != 0
    sc = (ctypes.c_double * 1)(1.0)
    kp = np.zeros((K, 2), np.float32); scr = np.zeros((K,), np.float32)
    rc = ctx.lib.sfd2_extract_multiscale(ctx.h, rgbx.ctypes.data, 0, h, w, sc, 1, ctypes.c_float(0.001), K, _lib.FLAG_IMG_U8_HWC | _lib.FLAG_IMG_U8_X,
                                         kp.ctypes.data, scr.ctypes.data, None, 0, K, None)
    assert rc != 0 and b"SFD2_FLAG_IMG_U8_X" in ctx.lib.sfd2_last_error()


def test_replica_lanes_carry_the_options_of_the_model(tmp_path, synth_sd):
    """Two lanes with an option that CHANGES the descriptors set on the model's context (comp_heads = 1), before and after the replica exists: every
    image gets the same result whichever context it lands on, i.e. the pipelined store equals the serial loop's."""
    from sfd2_amd import extract_localization as el
    model = _model(synth_sd, "f16c")
    items = [{"name": f"m/{i:02d}.png", "image": (synth.make_image(96, 128, 500 + i).transpose(1, 2, 0) * 255).astype(np.uint8), "original_size": (128, 96)}
             for i in range(10)]
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 120}}
    me = (model, el.extract_resnet_return)
    model.context.set_option("comp_heads", 1)
    a = el.main(conf, items, tmp_path / "s1", model_and_extractor=me, num_workers=0)
    b = el.main(conf, items, tmp_path / "p1", model_and_extractor=me, num_workers=2, lanes=2)      # the replica is made here, with the option
    _stores_equal(b, a)
    assert model.lanes(2)[1].context.options.get("comp_heads") == 1
    model.context.set_option("comp_heads", 0)                                                            # ... and changed after it exists
    c = el.main(conf, items, tmp_path / "s0", model_and_extractor=me, num_workers=0)
    d = el.main(conf, items, tmp_path / "p0", model_and_extractor=me, num_workers=2, lanes=2)
    _stores_equal(d, c)
    fa, fc = fio_open(a), fio_open(c)
    assert not np.array_equal(fa["m/00.png"]["descriptors"].__array__(), fc["m/00.png"]["descriptors"].__array__())   # (the option does change descriptors)


def test_replica_lanes_follow_new_weights_exponents_and_the_selfcheck(tmp_path, synth_sd):
    """cached replicas must not go stale.  (1) load_state_dict with other weights on a model whose lanes exist: the pipelined store (two lanes) still
    equals the serial loop's.  (2) calibrate_range on the model afterwards: the replica takes the new exponents.  (3) The replica RUNS with what the source's
    load-time self-check chose (sfd2_get_option), whatever the caller asked for before the load; range_status() covers every lane."""
    from sfd2_amd import extract_localization as el
    model = _model(synth_sd, "f16c")
    items = [{"name": f"m/{i:02d}.png", "image": (synth.make_image(96, 128, 700 + i).transpose(1, 2, 0) * 255).astype(np.uint8), "original_size": (128, 96)}
             for i in range(8)]
    name, conf = next(iter(el.confs.items()))
    conf = {**conf, "model": {**conf["model"], "max_keypoints": 120}}
    me = (model, el.extract_resnet_return)
    el.main(conf, items, tmp_path / "p0", model_and_extractor=me, num_workers=2, lanes=2)          # lanes exist now
    old_replica = model.lanes(2)[1]
    sd2 = synth.make_state_dict(3, family="student")                                                # (1) other weights (a family whose self-check moves options)
    model.load_state_dict(sd2)
    a = el.main(conf, items, tmp_path / "s1", model_and_extractor=me, num_workers=0)
    b = el.main(conf, items, tmp_path / "p1", model_and_extractor=me, num_workers=2, lanes=2)
    _stores_equal(b, a)
    rep = model.lanes(2)[1]
    assert rep is not old_replica
    for k in ("rb_inner", "comp_heads", "c3b_plain"):                                               # (3)
        assert rep.context.get_option(k) == model.context.get_option(k), k
    model.calibrate_range(synth.make_image(96, 128, 3) * 4.0)                                       # (2) other exponents
    e_src = model.context.act_exponents()[0]
    assert not np.array_equal(e_src, rep.context.act_exponents()[0])
    assert np.array_equal(model.lanes(2)[1].context.act_exponents()[0], e_src)
    c = el.main(conf, items, tmp_path / "s2", model_and_extractor=me, num_workers=0)
    d = el.main(conf, items, tmp_path / "p2", model_and_extractor=me, num_workers=2, lanes=2)
    _stores_equal(d, c)
    st = model.range_status()
    assert set(st["tensors"]) == set(rep.context.range_status()["tensors"]) and st["saturated"] == []
    for n, t in rep.context.range_status()["tensors"].items():
        assert st["tensors"][n]["max_stored"] >= t["max_stored"]


def fio_open(path):
    from sfd2_amd import feature_io as fio
    return fio.open_store(path, "r")


@pytest.mark.parametrize("flavour,mutual,ratio", [(0, 1, 0.0), (0, 0, 0.0), (0, 1, 0.9), (2, 1, 0.95)])
def test_match_batch_out16_equals_the_host_casts(flavour, mutual, ratio):
    """SFD2_FLAG_MATCH_OUT16: the matcher writes matches0 as int16 and scores0 as fp16 itself -- bit for bit what hloc/match_features.py:114,118's
    .short() / .half() make of the int64 / fp32 outputs (host and device outputs, a database set given by row selection)."""
    import torch
    ctx = _lib.default_context(0)
    rs = np.random.RandomState(7)
    def unit(n):
        d = rs.standard_normal((n, 128)).astype(np.float32)
        return d / np.linalg.norm(d, axis=1, keepdims=True)
    q = unit(700)
    dbs = [unit(n) for n in (512, 700, 33)]
    dbs[1][:50] = q[100:150] + 0.01 * rs.standard_normal((50, 128)).astype(np.float32)      # some real matches
    rows = np.arange(0, 700, 2, dtype=np.int32)
    qd = _lib.DescSet(q.ctypes.data, 700, _lib.DT_F32, _lib.LAYOUT_ND, 0, None, 0, 0)
    arr = (_lib.DescSet * 3)(_lib.DescSet(dbs[0].ctypes.data, 512, _lib.DT_F32, _lib.LAYOUT_ND, 0, None, 0, 0),
                             _lib.DescSet(dbs[1].ctypes.data, 700, _lib.DT_F32, _lib.LAYOUT_ND, 0, rows.ctypes.data, len(rows), 0),
                             _lib.DescSet(dbs[2].ctypes.data, 33, _lib.DT_F32, _lib.LAYOUT_ND, 0, None, 0, 0))
    conf = _lib.MatchConf(flavour, mutual, ratio, 0.0, _lib.SIM_F16)
    m64 = np.empty((3, 700), np.int64); s32 = np.empty((3, 700), np.float32)
    _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qd), arr, 3, 128, ctypes.byref(conf), m64.ctypes.data, s32.ctypes.data, 0, 0))
    m16 = np.empty((3, 700), np.int16); s16 = np.empty((3, 700), np.float16)
    _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qd), arr, 3, 128, ctypes.byref(conf), m16.ctypes.data, s16.ctypes.data, 0, _lib.FLAG_MATCH_OUT16))
    np.testing.assert_array_equal(m16, m64.astype(np.int16))
    np.testing.assert_array_equal(s16.view(np.uint16), s32.astype(np.float16).view(np.uint16))
    assert (m64 >= 0).sum() > 20
    md = torch.empty((3, 700), dtype=torch.int16, device="cuda"); sd = torch.empty((3, 700), dtype=torch.float16, device="cuda")
    _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qd), arr, 3, 128, ctypes.byref(conf), md.data_ptr(), sd.data_ptr(), 1, _lib.FLAG_MATCH_OUT16))
    np.testing.assert_array_equal(md.cpu().numpy(), m16)
    np.testing.assert_array_equal(sd.cpu().numpy().view(np.uint16), s16.view(np.uint16))


def test_resident_sets_prefetch_order_is_free(tmp_path):
    """ResidentSets with far more reads pending than pinned staging buffers, consumed in the REVERSE of the prefetch order: every set arrives
    (a read that finds no staging buffer goes through pageable memory; nothing waits for a buffer only the consumer can give back)."""
    import torch
    from sfd2_amd import feature_io as fio
    from sfd2_amd.pipeline import ResidentSets
    rs = np.random.RandomState(11)
    st = fio.open_store(str(tmp_path / "f.h5"), "w")
    names = [f"db/{i:03d}.jpg" for i in range(150)]
    ref = {}
    for nm in names:
        d = rs.standard_normal((128, 40)).astype(np.float64)
        ref[nm] = d
        st.write_group(nm, {"descriptors": d})
    st.close()
    feats = fio.open_store(st.path, "r")
    sets = ResidentSets(_lib.default_context(0), feats, readers=4)
    sets.prefetch(names)
    for nm in reversed(names):
        p, n = sets.get(nm, 0)
        assert n == 40 and p != 0
    _lib.default_context(0).sync()
    got = sets._sets[names[7]][0].cpu().numpy().reshape(40, 128)
    np.testing.assert_allclose(got.astype(np.float32), ref[names[7]].T.astype(np.float32), rtol=1e-3, atol=1e-6)      # (fp16 of the stored float64 set)
    assert sets.loads == 150
    sets.close()
