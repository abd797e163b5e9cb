#!/usr/bin/env python3
"""files -> feature store -> match store on one MI355X: the rate a user of the reference's two pipeline scripts sees.

    python tools/pipeline_bench.py [--queries 256 --db 128 --k 50 --workers 16 --size 1600x1200 --precision f16c,f16x3d,f16x3]

What it does (everything under a scratch directory, removed afterwards unless --keep):
  1. writes synthetic JPEGs (PIL encodes, quality 90, ~0.75 MB each -- the size of a real 1600x1200 photograph):
     `--queries` files under query/, `--db` under db/;
  2. extract_localization.main over all of them with `--workers` decoder threads (pipelined loop) -> feats-<conf> store:
     images/s, next to the decode-only ceiling of the same pool (no device work) and the serial loop on a subset;
  3. match_features.main (NNM) over queries x `--k` database images each (netvlad-style pair list) with the
     query-grouped, device-resident driver -> match store: pairs/s, next to the per-pair loop on a subset;
  4. checks on the subsets that the pipelined / grouped stores equal the serial ones dataset by dataset.
Prints one JSON line.  Weights are synth.make_state_dict (no checkpoint can travel); model construction, weight upload and
one warm-up image per geometry are outside the timed regions (a pipeline run over a dataset amortises them).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_images(root, n_query, n_db, H, W, quality=90, bases=4, seed=0):
    """n_query + n_db JPEG files; contents: `bases` synthetic images, lightly blurred (photograph-like spectrum, so the
    decoder's work per pixel is a photograph's), each file a different cyclic shift of one of them."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image, ImageFilter
    from sfd2_amd import synth
    os.makedirs(os.path.join(root, "query"), exist_ok=True)
    os.makedirs(os.path.join(root, "db"), exist_ok=True)
    base = []
    for b in range(bases):
        u8 = (synth.make_image(H, W, seed + b).transpose(1, 2, 0) * 255).astype(np.uint8)
        base.append(np.asarray(Image.fromarray(u8).filter(ImageFilter.GaussianBlur(0.8))))
    names = [f"query/q{i:05d}.jpg" for i in range(n_query)] + [f"db/d{i:05d}.jpg" for i in range(n_db)]
    rs = np.random.RandomState(seed)
    shifts = rs.randint(0, min(H, W), size=(len(names), 2))

    def enc(i):
        a = np.roll(base[i % bases], (int(shifts[i, 0]), int(shifts[i, 1])), axis=(0, 1))
        Image.fromarray(a).save(os.path.join(root, names[i]), format="JPEG", quality=quality)
        return os.path.getsize(os.path.join(root, names[i]))

    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        sizes = list(ex.map(enc, range(len(names))))
    return names, float(np.mean(sizes))


def decode_only_rate(ds, workers, rgbx=True):
    """The decoder pool alone (PIL decode + copy into a reusable buffer), no device: the ceiling files -> features can reach.
    rgbx: the pipelined loader's form (four-byte pixels pasted with the interpreter lock released); False: repacked to three bytes."""
    from sfd2_amd.pipeline import OrderedPrefetch
    class Buf:
        def __init__(self):
            self.a = np.empty(0, dtype=np.uint8)

        def reserve(self, n):
            if self.a.size < n:
                self.a = np.empty(n, dtype=np.uint8)
            return self.a[:n]

    import queue
    free = queue.Queue()
    for _ in range(workers + 4):
        free.put(Buf())

    def claim():
        try:
            return free.get(block=False)
        except queue.Empty:
            return None

    def load(idx, buf):
        return ds.load(idx, buf.reserve, rgbx=rgbx), buf

    pf = OrderedPrefetch(load, range(len(ds)), workers, window=workers + 2, claim=claim)
    t0 = time.perf_counter()
    n = 0
    while True:
        pf.top_up()
        if pf.ready:
            _, buf = pf.pop()
            free.put(buf)
            n += 1
        elif pf.exhausted:
            break
    dt = time.perf_counter() - t0
    pf.close()
    return n / dt


def stores_equal(a_path, b_path, keys=None):
    from sfd2_amd.feature_io import open_store
    a, b = open_store(a_path, "r"), open_store(b_path, "r")
    keys = list(b.keys()) if keys is None else keys
    for k in keys:
        for ds in b[k].keys():
            x, y = np.asarray(a[k][ds].__array__()), np.asarray(b[k][ds].__array__())
            if x.dtype != y.dtype or x.shape != y.shape or not np.array_equal(x, y):
                return False
    return len(keys) > 0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=256)
    ap.add_argument("--db", type=int, default=128)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--writers", type=int, default=3)
    ap.add_argument("--depth", type=int, default=6)
    ap.add_argument("--lanes", type=int, default=2, help="contexts (HIP streams) the pipelined extract loop alternates over")
    ap.add_argument("--size", default="1600x1200")
    ap.add_argument("--topk", type=int, default=4096)
    ap.add_argument("--precision", default="f16c,f16x3d,f16x3")
    ap.add_argument("--serial-images", type=int, default=32)
    ap.add_argument("--serial-pairs", type=int, default=200)
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--keep", action="store_true")
    return ap.parse_args(argv)


def run(args):
    """The whole measurement; returns the result dict (bench.py's `pipeline` object calls this with a smaller workload)."""
    W, H = (int(v) for v in args.size.split("x"))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("pipeline_bench needs an MI355X (there is no CPU path)")
    from sfd2_amd import extract_localization as el, match_features as mf, synth
    from sfd2_amd.model import ResSegNetV2

    scratch = args.scratch or tempfile.mkdtemp(prefix="sfd2_pipe_")
    os.makedirs(scratch, exist_ok=True)
    from sfd2_amd import feature_io as fio
    store0 = fio.STORE
    fio.STORE = "pack"       # the timed legs below write the fast stand-in store (as every round's numbers did); the `hdf5` object at the end is the same run into HDF5 files
    out = {"workload": f"{args.queries} query + {args.db} db JPEG files {W}x{H} -> top-{args.topk} features -> NNM matches, {args.k} db images per query",
           "decode_workers": args.workers, "writer_threads": args.writers, "extracts_in_flight": args.depth, "lanes": args.lanes,
           "host_logical_cpus": len(os.sched_getaffinity(0)), "host_cpu_count": os.cpu_count()}
    try:
        t0 = time.perf_counter()
        names, mean_bytes = write_images(os.path.join(scratch, "images"), args.queries, args.db, H, W)
        out["jpeg_mean_bytes"] = int(mean_bytes)
        out["encode_s"] = round(time.perf_counter() - t0, 2)
        conf_name = f"ressegnetv2-20220810-wapv2-sd2mfsf-uspg-0001-n{args.topk}-r1600"
        conf = el.confs[conf_name] if conf_name in el.confs else dict(next(iter(el.confs.values())))
        conf = {**conf, "model": {**conf["model"], "max_keypoints": args.topk}}
        ds = el.ImageDataset(os.path.join(scratch, "images"), conf["preprocessing"])
        assert len(ds) == len(names)
        out["decode_only_images_per_s"] = round(decode_only_rate(ds, args.workers), 1)
        out["decode_only_repacked_rgb_images_per_s"] = round(decode_only_rate(ds, args.workers, rgbx=False), 1)
        sd = synth.make_state_dict(0)
        sub_list = os.path.join(scratch, "subset.txt")
        with open(sub_list, "w") as f:
            f.write("\n".join(str(p) for p in ds.paths[:args.serial_images]) + "\n")
        ds_sub = el.ImageDataset(os.path.join(scratch, "images"), conf["preprocessing"], image_list=sub_list)
        feats_path = None
        for prec in args.precision.split(","):
            model = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
            model.load_state_dict(sd)
            model.cuda(0)
            me = (model, el.extract_resnet_return)
            for lm in model.lanes(args.lanes):                                                    # (the replicas of a multi-lane run: model construction, outside the timed region)
                el.extract_resnet_return(lm, ds[0]["image"], conf_th=0.001, topK=args.topk)       # workspace, first-use packing
            d_pipe, d_ser = os.path.join(scratch, f"pipe_{prec}"), os.path.join(scratch, f"serial_{prec}")
            t0 = time.perf_counter()
            p_pipe = el.main(conf, ds, d_pipe, model_and_extractor=me, num_workers=args.workers, writers=args.writers, depth=args.depth, lanes=args.lanes)
            dt = time.perf_counter() - t0
            t0 = time.perf_counter()
            p_ser = el.main(conf, ds_sub, d_ser, model_and_extractor=me, num_workers=0)
            dts = time.perf_counter() - t0
            from sfd2_amd.feature_io import open_store
            sub_keys = list(open_store(p_ser, "r").keys())
            out[f"extract_{prec}"] = {
                "files_to_features_images_per_s": round(len(ds) / dt, 1), "images": len(ds), "seconds": round(dt, 3),
                "serial_loop_images_per_s": round(len(ds_sub) / dts, 1), "serial_images": len(ds_sub),
                "equal_to_serial_on_subset": bool(stores_equal(p_pipe, p_ser, sub_keys)),
                "range_fallbacks": model.context.range_status()["fallbacks"],
                "mean_keypoints": float(np.mean([open_store(p_pipe, "r")[k]["scores"].shape[0] for k in sub_keys]))}
            if feats_path is None:
                feats_path, feats_dir = p_pipe, d_pipe
            del model
        # ---- matching: store -> store
        rs = np.random.RandomState(1)
        dbn = [n for n in names if n.startswith("db/")]
        qn = [n for n in names if n.startswith("query/")]
        k = min(args.k, len(dbn))
        pairs = [f"{q} {d}" for q in qn for d in rs.choice(dbn, size=k, replace=False)]
        mconf = mf.confs["NNM"]
        matcher = mf.load_matcher(mconf)
        matcher({"descriptors0": synth.make_descriptors(64, seed=1).T[None], "descriptors1": synth.make_descriptors(64, seed=2).T[None]})
        t0 = time.perf_counter()
        m_grp = mf.main(mconf, pairs, conf["output"], feats_dir, pairs_name="grouped", model=matcher, grouped=True)
        dtg = time.perf_counter() - t0
        sub_pairs = pairs[:args.serial_pairs]
        t0 = time.perf_counter()
        m_ser = mf.main(mconf, sub_pairs, conf["output"], feats_dir, pairs_name="serial", model=matcher, grouped=False)
        dtp = time.perf_counter() - t0
        # a second grouped pass over a fresh output with every set already read once by the OS (page cache): steady state of a long run
        t0 = time.perf_counter()
        mf.main(mconf, pairs, conf["output"], feats_dir, pairs_name="grouped2", model=matcher, grouped=True)
        dtg2 = time.perf_counter() - t0
        from sfd2_amd.feature_io import open_store
        out["match"] = {"store_to_store_pairs_per_s": round(len(pairs) / dtg, 1), "pairs": len(pairs), "seconds": round(dtg, 3),
                        "second_pass_pairs_per_s": round(len(pairs) / dtg2, 1),
                        "per_pair_loop_pairs_per_s": round(len(sub_pairs) / dtp, 1), "per_pair_pairs": len(sub_pairs),
                        "equal_to_per_pair_on_subset": bool(stores_equal(m_grp, m_ser, list(open_store(m_ser, "r").keys()))),
                        "k": k, "features": os.path.basename(feats_path)}
        # ---- the same two drivers writing HDF5 files directly (the reference's format; the default of open_store wherever an HDF5 library exists), and the
        # conversion of the stand-in stores afterwards (tools/pack_to_h5.py).  HDF5 costs ~250 us per pair group inside the library: the match store is format-bound.
        backend = fio.hdf5_backend()
        if backend is not None:
            fio.STORE = "h5"
            prec = args.precision.split(",")[0]
            model = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
            model.load_state_dict(sd)
            model.cuda(0)
            me = (model, el.extract_resnet_return)
            for lm in model.lanes(args.lanes):
                el.extract_resnet_return(lm, ds[0]["image"], conf_th=0.001, topK=args.topk)
            d_h5 = os.path.join(scratch, "h5")
            t0 = time.perf_counter()
            p_h5 = el.main(conf, ds, d_h5, model_and_extractor=me, num_workers=args.workers, writers=args.writers, depth=args.depth, lanes=args.lanes)
            dth = time.perf_counter() - t0
            n_pairs_h5 = min(len(pairs), 200 * k)      # (long enough to amortise the first read of every database set through HDF5)
            t0 = time.perf_counter()
            m_h5 = mf.main(mconf, pairs[:n_pairs_h5], conf["output"], d_h5, pairs_name="grouped", model=matcher, grouped=True)
            dtm = time.perf_counter() - t0
            eq_f = bool(stores_equal(p_h5, feats_path, list(fio.open_store(feats_path, "r").keys())[:16]))
            fio.STORE = "pack"
            t0 = time.perf_counter()
            n_f = fio.pack_to_h5(fio.open_store(feats_path, "r").path, os.path.join(scratch, "conv_feats.h5"), backend)
            dcf = time.perf_counter() - t0
            t0 = time.perf_counter()
            n_m = fio.pack_to_h5(fio.open_store(m_grp, "r").path, os.path.join(scratch, "conv_matches.h5"), backend)
            dcm = time.perf_counter() - t0
            out["hdf5"] = {"backend": backend.__name__, "files_to_features_images_per_s": round(len(ds) / dth, 1), "precision": prec,
                           "store_to_store_pairs_per_s": round(n_pairs_h5 / dtm, 1), "pairs": n_pairs_h5,
                           "features_equal_to_the_pack_run": eq_f, "is_hdf5": bool(open(p_h5, "rb").read(8) == b"\x89HDF\r\n\x1a\n"),
                           "pack_to_h5_feature_groups_per_s": round(n_f / dcf, 1), "pack_to_h5_pair_groups_per_s": round(n_m / dcm, 1)}
            del model
    finally:
        fio.STORE = store0
        if not args.keep:
            shutil.rmtree(scratch, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(json.dumps(run(parse_args())))
