"""CPU tests of the pipelined drivers' host side (no GPU: stub extractor): the stand-in stores, the ordered decode pool,
the writer pool, and that extract_localization.main writes the same store with num_workers > 0 as with the serial loop."""
import json
import os
import threading
import time

import numpy as np
import pytest

from sfd2_amd import feature_io as fio


def test_pack_store_layout_roundtrip_and_reopen(tmp_path, monkeypatch):
    p = str(tmp_path / "feats-x.h5")
    monkeypatch.setattr(fio, "STORE", "pack")          # the stand-in (the default is a real HDF5 file wherever an HDF5 library exists: test_h5_store_* below)
    st = fio.open_store(p, "a")
    assert isinstance(st, fio.PackStore) and st.path.endswith("feats-x.pack")
    g = st.create_group("db/1.jpg")
    g.create_dataset("keypoints", data=np.arange(8.).reshape(4, 2))
    g.create_dataset("image_size", data=np.array([640, 480]))
    assert "keypoints" in g and sorted(g.keys()) == ["image_size", "keypoints"]
    st.write_group("query/2.jpg", {"descriptors": np.ones((128, 3)), "scores": np.zeros((0,)), "h": np.float16([1.5, 2.5])})
    with pytest.raises(ValueError, match="already exists"):
        st.create_group("db/1.jpg")
    with pytest.raises(ValueError, match="already exists"):
        st["db/1.jpg"].create_dataset("keypoints", data=np.zeros(1))
    # readable while open for writing (the drivers' `pair in store` / resume checks)
    np.testing.assert_array_equal(st["db/1.jpg"]["keypoints"].__array__(), np.arange(8.).reshape(4, 2))
    st.close()
    rd = fio.open_store(p, "r")
    assert rd.keys() == ["db/1.jpg", "query/2.jpg"] and "db/1.jpg" in rd and "nope" not in rd
    with pytest.raises(KeyError):
        rd["nope"]
    g = rd["query/2.jpg"]
    assert g["descriptors"].shape == (128, 3) and g["descriptors"].dtype == np.float64
    assert g["scores"].shape == (0,) and g["h"].dtype == np.float16 and g["h"][1] == 2.5
    assert rd["db/1.jpg"]["image_size"][()].tolist() == [640, 480]
    with pytest.raises(IOError):
        rd.write_group("x", {})
    rd.close()
    # append to an existing store; a torn index line (killed writer) is ignored and overwritten data is never referenced
    with open(os.path.join(st.path, "index.jsonl"), "a") as f:
        f.write('{"g":"torn","d":{"a":["<f8",[2],')
    with open(os.path.join(st.path, "data.bin"), "ab") as f:
        f.write(b"xyz")
    ap = fio.open_store(p, "a")
    assert "torn" not in ap
    ap.write_group("later", {"v": np.arange(5, dtype=np.int16)})
    ap.close()
    rd = fio.open_store(p, "r")
    np.testing.assert_array_equal(rd["later"]["v"].__array__(), np.arange(5, dtype=np.int16))
    assert rd["query/2.jpg"]["h"][0] == 1.5
    # mode 'w' starts empty
    assert fio.open_store(p, "w").keys() == []


@pytest.mark.parametrize("kind", ["pack", "h5"])
def test_store_concurrent_writers_and_readers(tmp_path, kind):
    if kind == "h5" and fio.hdf5_backend() is None:
        pytest.skip("no HDF5 library on this host")
    st = fio.open_store(str(tmp_path / "m.h5"), "w", standin=kind)
    assert isinstance(st, fio.PackStore if kind == "pack" else fio.H5Store)

    def work(t):
        for i in range(200):
            st.write_group(f"t{t}_{i}", {"matches0": np.full(64, t * 1000 + i, dtype=np.int16), "s": np.full(3, i, dtype=np.float16)})
            if i % 50 == 49:
                assert st[f"t{t}_{i - 7}"]["matches0"][0] == t * 1000 + i - 7
    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    st.close()
    rd = fio.open_store(str(tmp_path / "m.h5"), "r")
    assert len(rd.keys()) == 800
    for t in range(4):
        for i in (0, 57, 199):
            assert (rd[f"t{t}_{i}"]["matches0"].__array__() == t * 1000 + i).all()


def test_open_store_reads_round4_npz_shards(tmp_path):
    old = fio.open_store(str(tmp_path / "f.h5"), "w", standin="npz")
    fio.write_features(old, "a/b.jpg", {"keypoints": np.zeros((2, 2)), "image_size": np.array([3, 4])})
    old.close()
    rd = fio.open_store(str(tmp_path / "f.h5"), "r")
    assert isinstance(rd, fio.NpzStore) and rd.keys() == ["a/b.jpg"]
    with pytest.raises(ValueError):
        fio.open_store(str(tmp_path / "g.h5"), "w", standin="zip")


def test_ordered_prefetch_keeps_order_and_deals_buffers_in_order():
    from sfd2_amd.pipeline import OrderedPrefetch
    import queue
    free = queue.Queue()
    for b in range(3):
        free.put(b)
    claimed = []

    def claim():
        try:
            b = free.get(block=False)
        except queue.Empty:
            return None
        claimed.append(b)
        return b

    def load(idx, buf):
        time.sleep(0.002 * ((idx * 7) % 5))      # later items often finish first
        return idx, buf

    pf = OrderedPrefetch(load, range(40), workers=4, window=6, claim=claim)
    got = []
    while True:
        pf.top_up()
        if pf.ready:
            idx, buf = pf.pop()
            got.append(idx)
            free.put(buf)                        # the consumer gives the buffer back: with 3 buffers the pool never runs further ahead
        elif pf.exhausted:
            break
    pf.close()
    assert got == list(range(40)) and len(claimed) == 40
    # no claim: plain windowed prefetch
    pf = OrderedPrefetch(lambda i, r: i * i, range(10), workers=3, window=4)
    out = []
    while True:
        pf.top_up()
        if pf.ready:
            out.append(pf.pop())
        elif pf.exhausted:
            break
    pf.close()
    assert out == [i * i for i in range(10)]


def test_writer_pool_surfaces_errors():
    from sfd2_amd.pipeline import WriterPool
    seen = []

    def fn(job):
        if job == 3:
            raise RuntimeError("disk full")
        seen.append(job)
    wp = WriterPool(fn, workers=2)
    for j in range(6):
        try:
            wp.put(j)
        except RuntimeError:
            break
    with pytest.raises(RuntimeError, match="disk full"):
        wp.close()
    assert 3 not in seen


def test_writer_pool_error_does_not_strand_a_producer_waiting_for_its_resources():
    """the producers of both drivers wait for a RESOURCE the writer gives back (an extractor slot, a ring of pinned buffers), not for the
    queue.  After a failed write the queued jobs are dropped through on_drop, which returns what they carry: the producer wakes up and meets the
    error in its next put() instead of hanging."""
    import queue
    from sfd2_amd.pipeline import WriterPool
    free = queue.Queue()
    for r in range(3):
        free.put(r)
    gate = threading.Event()

    def fn(job):
        gate.wait(5)                    # a slow writer: the producer gets ahead and runs out of rings
        try:
            if job[1] == 0:
                raise OSError("No space left on device")
        finally:
            free.put(job[0])

    wp = WriterPool(fn, workers=1, maxsize=2, on_drop=lambda job: free.put(job[0]))
    out = {}

    def producer():
        try:
            for j in range(50):
                ring = free.get(timeout=10)         # (the drivers wait without a timeout; the test must not hang if the fix regresses)
                if j == 2:
                    gate.set()
                out["held"] = ring
                wp.put((ring, j))
            out["err"] = None
        except BaseException as e:                  # noqa: BLE001
            out["err"] = e
            free.put(out["held"])                   # the ring in the producer's hand when put() raised was never queued
    t = threading.Thread(target=producer)
    t.start()
    t.join(20)
    assert not t.is_alive(), "producer stranded"
    assert isinstance(out["err"], OSError), out["err"]
    with pytest.raises(OSError):
        wp.close()
    assert sorted(free.queue) == [0, 1, 2]          # every ring came back exactly once


def _stub_extractor(model, img, topK, mask, conf_th, scales):
    a = np.asarray(img, dtype=np.float64).reshape(-1)
    n = 5 + int(a[:7].sum()) % 11
    rs = np.random.RandomState(int(a[:16].sum() * 1000) % (2 ** 31))
    return {"keypoints": rs.randint(0, 100, size=(n, 2)).astype(np.float64), "descriptors": rs.rand(n, 128),
            "scores": np.sort(rs.rand(n))[::-1].copy()}


def test_pipelined_main_equals_serial_main_with_stub_extractor(tmp_path):
    """extract_localization.main(num_workers=3) -- decode pool, synchronous stub extractor, writer threads -- writes the
    store the serial loop writes; with the tag filter; and sharded over two ranks + merge."""
    from sfd2_amd import extract_localization as el
    rs = np.random.RandomState(0)
    items = [{"name": f"{'query' if i % 4 == 0 else 'db'}/im{i:03d}.jpg", "image": rs.rand(3, 24, 32).astype(np.float32),
              "original_size": (64, 48)} for i in range(23)]
    name, conf = next(iter(el.confs.items()))
    me = (None, _stub_extractor)

    def equal(a, b):
        x, y = fio.open_store(a, "r"), fio.open_store(b, "r")
        assert list(x.keys()) == list(y.keys()) and len(list(x.keys())) > 0
        for k in x.keys():
            for ds in y[k].keys():
                u, v = np.asarray(x[k][ds].__array__()), np.asarray(y[k][ds].__array__())
                assert u.dtype == v.dtype and np.array_equal(u, v), (k, ds)
        return list(x.keys())

    a = el.main(conf, items, tmp_path / "s", model_and_extractor=me, num_workers=0)
    b = el.main(conf, items, tmp_path / "p", model_and_extractor=me, num_workers=3, writers=2)
    assert len(equal(b, a)) == 23
    at = el.main(conf, items, tmp_path / "st", model_and_extractor=me, tag="query", num_workers=0)
    bt = el.main(conf, items, tmp_path / "pt", model_and_extractor=me, tag="query", num_workers=2)
    assert len(equal(bt, at)) == 6
    # two ranks in turn (no barrier needed in one process), pipelined, then rank 0's merge
    el.main(conf, items, tmp_path / "w", model_and_extractor=me, world=2, rank=1, num_workers=2)
    m = el.main(conf, items, tmp_path / "w", model_and_extractor=me, world=2, rank=0, num_workers=2)
    assert len(equal(m, a)) == 23
    idx = json.load(open(str(tmp_path / "w" / (conf["output"] + ".part1of2.h5.index.json"))))
    assert [i for i, _ in idx] == list(range(1, 23, 2))


def test_group_pairs_is_query_major_in_first_appearance_order():
    from sfd2_amd import match_features as mf
    pairs = mf.unique_pairs(["q1 a", "q0 a", "q1 b", "a q1", "q0 c", "q1 c", "q0 a"])
    assert pairs == [("q1", "a"), ("q0", "a"), ("q1", "b"), ("q0", "c"), ("q1", "c")]
    assert mf.group_pairs(pairs) == [("q1", [(0, "a"), (2, "b"), (4, "c")]), ("q0", [(1, "a"), (3, "c")])]


def test_pipelined_main_surfaces_decode_errors(tmp_path):
    """An unreadable image raises in the pipelined loop as it does in the serial one (extract_localization.py:166-167), it does not hang
    the pool; groups written before it stay readable."""
    from sfd2_amd import extract_localization as el

    class Items:
        def __len__(self):
            return 9

        def __getitem__(self, idx):
            if idx == 5:
                raise ValueError("Cannot read image broken.jpg.")
            return {"name": f"im{idx}.jpg", "image": np.full((3, 8, 8), idx, np.float32), "original_size": (8, 8)}

    name, conf = next(iter(el.confs.items()))
    for nw in (0, 3):
        with pytest.raises(ValueError, match="Cannot read image"):
            el.main(conf, Items(), tmp_path / f"o{nw}", model_and_extractor=(None, _stub_extractor), num_workers=nw)
        st = fio.open_store(str(tmp_path / f"o{nw}" / (conf["output"] + ".h5")), "r")
        assert set(st.keys()) <= {f"im{i}.jpg" for i in range(5)} and len(st.keys()) >= (5 if nw == 0 else 0)


def test_decoder_rgbx_paste_equals_repacked_pixels(tmp_path):
    """_read_rgb_u8(rgbx=True): PIL's four-byte pixels pasted into the caller's buffer with the interpreter lock released -- the first three bytes of every
    pixel are what np.asarray(im) gives (the path the serial loop takes); odd sizes and a non-RGB file included."""
    PIL = pytest.importorskip("PIL")
    from PIL import Image
    from sfd2_amd import extract_localization as el
    rs = np.random.RandomState(3)
    for i, (h, w, mode) in enumerate([(48, 64, "RGB"), (37, 53, "RGB"), (40, 40, "L")]):
        arr = rs.randint(0, 256, (h, w, 3) if mode == "RGB" else (h, w), dtype=np.uint8)
        p = tmp_path / f"im{i}.png"
        Image.fromarray(arr, mode).save(p)
        want = el._read_rgb_u8(p)
        store = {}
        def reserve(n):
            store["b"] = np.zeros(n + 7, np.uint8)[:n]
            return store["b"]
        got = el._read_rgb_u8(p, reserve, rgbx=True)
        assert got.dtype == np.uint8 and got.shape[:2] == (h, w) and got.shape[2] in (3, 4)
        np.testing.assert_array_equal(got[:, :, :3], want)
        assert got.base is not None and np.shares_memory(got, store["b"])          # the pixels are in the caller's buffer
    assert el._RGBX_OK in (True, False)


def test_packstore_write_rows_equals_single_appends(tmp_path):
    """PackStore.write_rows (a query's pair groups as ONE append) against write_group per pair: the same groups, dtypes, shapes and bytes; rows whose
    size is not a multiple of 64 bytes take the per-row path; existing / duplicate names are refused before anything is written."""
    from sfd2_amd import feature_io as fio
    rs = np.random.RandomState(5)
    for n in (4096, 100):                  # 8 KB rows (aligned) / 200-byte rows (not)
        m = rs.randint(-1, n, (7, n)).astype(np.int16); s = rs.rand(7, n).astype(np.float16)
        a = fio.open_store(str(tmp_path / f"a{n}.h5"), "w"); b = fio.open_store(str(tmp_path / f"b{n}.h5"), "w")
        names = [f"q_d{i}" for i in range(7)]
        a.write_rows(names, {"matches0": m, "matching_scores0": s})
        for i, nm in enumerate(names):
            b.write_group(nm, {"matches0": m[i], "matching_scores0": s[i]})
        with pytest.raises(ValueError):
            a.write_rows(["q_d3", "new"], {"matches0": m[:2], "matching_scores0": s[:2]})
        if n == 4096:
            with pytest.raises(ValueError):
                a.write_rows(["x", "x"], {"matches0": m[:2], "matching_scores0": s[:2]})
        assert "new" not in a and "x" not in a
        a.close(); b.close()
        ra, rb = fio.open_store(a.path, "r"), fio.open_store(b.path, "r")
        assert list(ra.keys()) == list(rb.keys()) == names
        for nm in names:
            for k in ("matches0", "matching_scores0"):
                x, y = ra[nm][k], rb[nm][k]
                assert x.dtype == y.dtype and x.shape == y.shape
                np.testing.assert_array_equal(x[()], y[()])


def test_packstore_survives_a_writer_killed_mid_run(tmp_path):
    """the data file is flushed before an index line is written, so a snapshot of the two files taken while a writer runs (= what a
    killed process leaves) never holds a line whose bytes are missing; and a store whose index DOES run ahead of its data (written here by hand: the
    state the unflushed writer could leave) is cut at the first such line on open -- those groups read as absent, not as later appends' bytes."""
    import shutil
    p = str(tmp_path / "k.h5")
    st = fio.open_store(p, "w", standin="pack")
    snaps = []
    for i in range(400):
        st.write_group(f"g{i}", {"v": np.full(5, i, dtype=np.int16)})
        if i % 57 == 56:                       # what is on disk NOW, without a flush() or close() by the writer
            d = str(tmp_path / f"snap{i}.pack"); os.makedirs(d)
            # index first: a line on disk must name bytes that are on disk already
            shutil.copy(os.path.join(st.path, "index.jsonl"), d); shutil.copy(os.path.join(st.path, "data.bin"), d)
            snaps.append(d)
    st.close()
    for d in snaps:
        rd = fio.PackStore(d, "r")
        for nm in rd.keys():
            assert rd[nm]["v"][()].tolist() == [int(nm[1:])] * 5
        rd.close()
    # an index that runs ahead of the data: cut, then appended to -- the dropped names stay absent
    d = str(tmp_path / "ahead.pack"); shutil.copytree(st.path, d)
    with open(os.path.join(d, "data.bin"), "r+b") as f:
        f.truncate(64 * 388 + 3)               # group 388's bytes are torn, 389.. are gone
    ap = fio.PackStore(d, "a")
    assert "g387" in ap and "g388" not in ap and "g399" not in ap and len(ap.keys()) == 388
    ap.write_group("new", {"v": np.full(5, -7, dtype=np.int16)})
    ap.close()
    rd = fio.PackStore(d, "r")
    assert rd["new"]["v"][()].tolist() == [-7] * 5 and rd["g387"]["v"][()].tolist() == [387] * 5 and "g390" not in rd
    with open(os.path.join(d, "index.jsonl")) as f:
        assert len(f.readlines()) == 389


def test_decoder_prefers_cv2_when_importable(tmp_path, monkeypatch):
    """The reference decodes with cv2.imread (extract_localization.py:162-165); _read_rgb_u8 takes that decoder wherever it is importable and PIL otherwise
    (VERDICT r5 #7).  cv2 is not in this image: a stand-in module with the three entry points used (imread / cvtColor / the constants) checks the branch --
    same pixels, three- and four-byte forms, the caller's buffer, the ValueError of :166-167."""
    import sys
    import types
    from PIL import Image
    from sfd2_amd import extract_localization as el
    rs = np.random.RandomState(3)
    rgb = rs.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    calls = []
    fake = types.ModuleType("cv2")
    fake.IMREAD_COLOR, fake.COLOR_BGR2RGB, fake.COLOR_BGR2RGBA = 1, 4, 2

    def imread(path, flag):
        calls.append(path)
        try:
            return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
        except OSError:
            return None

    def cvt(img, code, dst=None):
        res = img[:, :, ::-1] if code == 4 else np.concatenate([img[:, :, ::-1], np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)
        if dst is not None and code == 4:
            np.copyto(dst, res)
            return dst
        return np.ascontiguousarray(res)                                # (the four-byte form ignores dst: the copy path)
    fake.imread, fake.cvtColor = imread, cvt
    monkeypatch.setitem(sys.modules, "cv2", fake)
    monkeypatch.setattr(el, "_CV2", False)
    monkeypatch.delenv("SFD2_DECODER", raising=False)
    try:
        np.testing.assert_array_equal(el._read_rgb_u8(tmp_path / "a.png"), rgb)
        buf = np.zeros(20 * 30 * 4, np.uint8)
        out = el._read_rgb_u8(tmp_path / "a.png", reserve=lambda n: buf[:n])
        assert np.shares_memory(out, buf) and out.shape == (20, 30, 3)
        np.testing.assert_array_equal(out, rgb)
        out4 = el._read_rgb_u8(tmp_path / "a.png", reserve=lambda n: buf[:n], rgbx=True)
        assert np.shares_memory(out4, buf) and out4.shape == (20, 30, 4)
        np.testing.assert_array_equal(out4[:, :, :3], rgb)
        assert len(calls) == 3
        (tmp_path / "bad.jpg").write_bytes(b"not an image")
        with pytest.raises(ValueError, match="Cannot read image"):
            el._read_rgb_u8(tmp_path / "bad.jpg")
        monkeypatch.setattr(el, "_CV2", False)
        monkeypatch.setenv("SFD2_DECODER", "pil")                       # forced back to PIL
        np.testing.assert_array_equal(el._read_rgb_u8(tmp_path / "a.png"), rgb)
        assert len(calls) == 4
    finally:
        el._CV2 = False


def test_h5_store_is_a_real_hdf5_file_with_the_reference_layout(tmp_path):
    """VERDICT r5 'what's missing' 2 / SURVEY 8f1: the stores ARE HDF5 files wherever an HDF5 library exists -- h5py, or the C library itself through
    sfd2_amd/h5lite.py (this image has libhdf5 and no h5py).  Written through the stores' interface, then checked THREE ways: read back through the store,
    through a second independent open of the file, and by HDF5's own h5dump (group nesting along '/', dataset names, shapes and types of
    extract_localization.py:266-272 and hloc/match_features.py:108-119)."""
    import shutil
    import subprocess
    backend = fio.hdf5_backend()
    if backend is None:
        pytest.skip("no HDF5 library on this host")
    rs = np.random.RandomState(0)
    pred = {"keypoints": rs.random_sample((17, 2)), "descriptors": rs.random_sample((128, 17)), "scores": rs.random_sample(17),
            "image_size": np.array([160, 120])}
    p = str(tmp_path / "feats.h5")
    with fio.open_store(p, "w") as st:
        assert isinstance(st, fio.H5Store)
        fio.write_features(st, "db/1.jpg", pred)
        fio.write_features(st, "query/day/2.jpg", {**pred, "scores": np.zeros((0,))})
        with pytest.raises(ValueError, match="already exists"):
            st.create_group("db/1.jpg")
        assert "db/1.jpg" in st and "db/9.jpg" not in st and "nope/x" not in st
        assert st.keys() == ["db/1.jpg", "query/day/2.jpg"]
    assert os.path.isfile(p)
    with open(p, "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"                               # the HDF5 signature
    rd = fio.open_store(p, "r")
    assert isinstance(rd, fio.H5Store) and rd.keys() == ["db/1.jpg", "query/day/2.jpg"]
    for k, v in pred.items():
        got = rd["db/1.jpg"][k].__array__()
        assert got.dtype == np.asarray(v).dtype and np.array_equal(got, v), k
    assert rd["query/day/2.jpg"]["scores"].shape == (0,)
    with pytest.raises(KeyError):
        rd["db/2.jpg"]
    with pytest.raises(IOError):
        rd.write_group("x", {"v": np.zeros(2)})
    rd.close()
    # matches: int16 / fp16 as stored by the reference (hloc/match_features.py:114,118)
    m = np.arange(-1, 16).astype(np.int64)
    sc = rs.random_sample(17).astype(np.float32)
    q = str(tmp_path / "matches.h5")
    with fio.open_store(q, "w") as st:
        fio.write_matches(st, "query-q1.jpg_db-1.jpg", m, sc)
        st.write_rows(["a_b", "c_d"], {"matches0": np.full((2, 64), 7, np.int16), "matching_scores0": np.full((2, 64), 0.5, np.float16)})
    with fio.open_store(q, "a") as st:                                          # append to an existing file
        st.write_group("e_f", {"matches0": np.zeros(3, np.int16), "matching_scores0": np.zeros(3, np.float16)})
        g = st["query-q1.jpg_db-1.jpg"]
        assert g["matches0"].dtype == np.int16 and g["matching_scores0"].dtype == np.float16
        np.testing.assert_array_equal(g["matches0"][()], m.astype(np.int16))
        np.testing.assert_array_equal(g["matching_scores0"][()], sc.astype(np.float16))
        assert st.keys() == ["a_b", "c_d", "e_f", "query-q1.jpg_db-1.jpg"] and st["c_d"]["matches0"][5] == 7
    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump:
        out = subprocess.run([h5dump, "-H", p], capture_output=True, text=True, check=True).stdout
        flat = " ".join(out.split())
        assert 'GROUP "db" { GROUP "1.jpg" {' in flat and 'GROUP "query" { GROUP "day" { GROUP "2.jpg" {' in flat
        assert 'DATASET "descriptors" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 128, 17 ) / ( 128, 17 ) }' in flat
        assert 'DATASET "keypoints" { DATATYPE H5T_IEEE_F64LE DATASPACE SIMPLE { ( 17, 2 ) / ( 17, 2 ) }' in flat
        assert 'DATASET "image_size" { DATATYPE H5T_STD_I64LE DATASPACE SIMPLE { ( 2 ) / ( 2 ) }' in flat
        out = " ".join(subprocess.run([h5dump, "-H", q], capture_output=True, text=True, check=True).stdout.split())
        assert 'DATASET "matches0" { DATATYPE H5T_STD_I16LE DATASPACE SIMPLE { ( 17 ) / ( 17 ) }' in out
        assert 'DATASET "matching_scores0" { DATATYPE 16-bit little-endian floating-point' in out
        data = subprocess.run([h5dump, "-d", "/query-q1.jpg_db-1.jpg/matches0", q], capture_output=True, text=True, check=True).stdout
        assert "-1, 0, 1, 2, 3" in data


def test_pack_to_h5_hands_a_fast_store_over_to_the_reference_format(tmp_path):
    """A run that wrote the PackStore stand-in (SFD2_STORE=pack: the pipelined match driver outruns HDF5's ~4 k pair groups/s) is converted afterwards --
    feature_io.pack_to_h5 / tools/pack_to_h5.py: every group, dataset, dtype and byte; reading prefers the HDF5 file once it exists."""
    import subprocess
    import sys
    if fio.hdf5_backend() is None:
        pytest.skip("no HDF5 library on this host")
    rs = np.random.RandomState(1)
    src = fio.open_store(str(tmp_path / "feats.h5"), "w", standin="pack")
    want = {}
    for i in range(20):
        n = int(rs.randint(0, 40))
        want[f"db/seq{i % 3}/{i}.jpg"] = {"keypoints": rs.rand(n, 2), "descriptors": rs.rand(128, n), "scores": rs.rand(n), "image_size": np.array([640 + i, 480])}
        src.write_group(f"db/seq{i % 3}/{i}.jpg", want[f"db/seq{i % 3}/{i}.jpg"])
    src.write_rows([f"q_{j}" for j in range(4)], {"matches0": rs.randint(-1, 99, (4, 64)).astype(np.int16), "matching_scores0": rs.rand(4, 64).astype(np.float16)})
    src.close()
    assert isinstance(fio.open_store(str(tmp_path / "feats.h5"), "r"), fio.PackStore)          # only the stand-in exists so far
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pack_to_h5.py")
    r = subprocess.run([sys.executable, tool, src.path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "24 groups" in r.stdout
    rd = fio.open_store(str(tmp_path / "feats.h5"), "r")
    assert isinstance(rd, fio.H5Store) and len(rd.keys()) == 24
    pk = fio.PackStore(src.path, "r")
    for name in pk.keys():
        assert sorted(pk[name].keys()) == sorted(rd[name].keys())
        for k in pk[name].keys():
            a, b = pk[name][k].__array__(), rd[name][k].__array__()
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), (name, k)
