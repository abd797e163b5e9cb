#!/usr/bin/env python3
"""Can ONE host feed eight GPUs?  The host side of an 8-GPU run, measured on whatever box this runs on (VERDICT r5 #5b).

    python tools/host_soak.py --ranks 8 [--workers 16 --writers 2 --seconds 15 --store pack|h5 --no-affinity --per-rank-device-rate 715]

`--ranks` processes run at the same time, each the host half of one rank of the pipelined drivers with the device stage stubbed out:

  extract   `--workers` decoder threads (sfd2_amd.extract_localization.ImageDataset.load: the decoder, four-byte pixels into reusable buffers, in item
            order through pipeline.OrderedPrefetch) -> [device stage: none; the slot's float32 results are a fixed template] -> `--writers` writer threads
            doing what _extract_pipelined's writers do per image (float32 -> float64, key-point rescale, transpose, one feature group of 4096 key points
            = 4.3 MB into the rank's store);
  match     one producer handing a query's [50, 4096] int16 / fp16 blocks (what sfd2_match_batch writes with SFD2_FLAG_MATCH_OUT16) to the match driver's
            writer (one write_rows per query), plus `--readers` threads reading float64 descriptor sets from a feature store into buffers (ResidentSets'
            reader threads; in a real run each database set is read once and then lives in HBM).

The reference's shape is one process per GPU with 4 DataLoader workers (extract_localization.py:230-240) and a per-pair loop
(hloc/match_features.py:99-119).  Placement: rank r is pinned as sharding.share_of_cpus would pin it on a node with `ranks / 2` GPUs per socket
(the two halves of the rank list on NUMA node 0 / 1; this box has one GPU, so the PCI lookup itself is exercised by tests/test_cpu_placement.py).
Prints one JSON object: per-rank and aggregate decoded-and-stored images/s and stored pairs/s, against ranks x the per-rank device rate."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def node_cpulists():
    from sfd2_amd.sharding import parse_cpulist
    out = []
    base = "/sys/devices/system/node"
    try:
        nodes = sorted(int(n[4:]) for n in os.listdir(base) if n.startswith("node") and n[4:].isdigit())
    except OSError:
        nodes = []
    for n in nodes:
        try:
            with open(f"{base}/node{n}/cpulist") as f:
                cpus = parse_cpulist(f.read())
            if cpus:
                out.append(cpus)
        except OSError:
            pass
    return out or [sorted(os.sched_getaffinity(0))]


def child(args):
    from sfd2_amd import feature_io as fio
    from sfd2_amd.extract_localization import ImageDataset, rescale_keypoints
    from sfd2_amd.pipeline import OrderedPrefetch, WriterPool
    from sfd2_amd.sharding import share_of_cpus
    r, n = args.child, args.ranks
    placement = {"pinned": False}
    if not args.no_affinity and hasattr(os, "sched_setaffinity"):
        nodes = node_cpulists()
        per = max(1, n // len(nodes))
        lists = [nodes[min(i // per, len(nodes) - 1)] for i in range(n)]
        cpus = share_of_cpus(lists, r, allowed=sorted(os.sched_getaffinity(0)))
        if cpus:
            os.sched_setaffinity(0, cpus)
            placement = {"pinned": True, "cpus": len(cpus), "first": cpus[0], "last": cpus[-1]}
    fio.STORE = args.store
    ds = ImageDataset(args.images, {"resize_max": 1600})
    K, TOPK = args.k, 4096
    rs = np.random.RandomState(r)
    t_kp = (rs.rand(TOPK, 2) * 1600).astype(np.float32)
    t_sc = np.sort(rs.rand(TOPK).astype(np.float32))[::-1].copy()
    t_de = rs.randn(TOPK, 128).astype(np.float32)
    t_de64 = np.ascontiguousarray(t_de.T, dtype=np.float64)
    scratch = os.path.join(args.scratch, f"rank{r}")
    os.makedirs(scratch, exist_ok=True)
    # Bounded footprint: at the device rate a rank writes 3 GB/s of float64 feature groups -- 15 s of eight ranks would be 360 GB.  Every writer thread
    # owns a store and starts it over (mode "w") every `recycle` groups / queries: at most ~0.25 GB per thread on the scratch file system at any time.
    tls = threading.local()
    all_stores = []

    def my_store(kind, recycle):
        st = getattr(tls, kind, None)
        n = getattr(tls, kind + "_n", 0)
        if st is None or n >= recycle:
            if st is not None:
                st.close()
            st = fio.open_store(os.path.join(scratch, f"{kind}-{threading.get_ident()}.h5"), "w")
            setattr(tls, kind, st)
            all_stores.append(st)
            n = 0
        setattr(tls, kind + "_n", n + 1)
        return st
    stop_at = [None]
    counts = {"images": 0, "pairs": 0, "sets_read": 0}
    lock = threading.Lock()

    # ---- extract half
    class Buf:
        def __init__(self):
            self.a = np.empty(0, dtype=np.uint8)

        def reserve(self, nb):
            if self.a.size < nb:
                self.a = np.empty(nb, dtype=np.uint8)
            return self.a[:nb]

    import queue
    free = queue.Queue()
    for _ in range(args.workers + 6):
        free.put(Buf())

    def claim():
        try:
            return free.get(block=False)
        except queue.Empty:
            return None

    def load(idx, buf):
        return ds.load(idx % len(ds), buf.reserve, rgbx=True), buf

    def write(job):
        i, data = job
        if args.host_cast:       # rounds 1-5: the writer casts float32 [n][128] to float64 and the store's copy transposes it
            kp, sc, de = t_kp.astype(np.float64), t_sc.astype(np.float64), t_de.astype(np.float64)
        else:                    # round 6 (SFD2_FLAG_DESC_STORE64): the slot already holds float64 [128][n]; slot_arrays hands out its transposed view
            kp, sc, de = t_kp.astype(np.float64), t_sc.astype(np.float64), t_de64.T
        pred = {"keypoints": rescale_keypoints(kp, data["original_size"], np.array((1600, 1200))), "descriptors": de.transpose(), "scores": sc,
                "image_size": data["original_size"]}
        fio.write_features(my_store("feats", 48), f"query/r{r}_{i:07d}.jpg", pred)
        with lock:
            counts["images"] += 1

    def extract_half():
        def indices():
            i = r
            while time.perf_counter() < stop_at[0]:
                yield i
                i += n
        pf = OrderedPrefetch(load, indices(), args.workers, window=args.workers + 2, claim=claim)
        wp = WriterPool(write, workers=args.writers, maxsize=4 * args.writers)
        k = 0
        try:
            while True:
                pf.top_up()
                if pf.ready:
                    data, buf = pf.pop()
                    # (the device stage would read the pixels here: upload + extract; the buffer goes back as the real loop returns it after the upload)
                    meta = {"original_size": data["original_size"]}
                    free.put(buf)
                    if args.half == "decode":          # the decoder pool alone: no float64 groups, no store
                        with lock:
                            counts["images"] += 1
                    else:
                        wp.put((k, meta))
                    k += 1
                elif pf.exhausted:
                    break
        finally:
            pf.close()
            wp.close()

    # ---- match half
    m_blk = rs.randint(-1, TOPK, (K, TOPK)).astype(np.int16)
    s_blk = rs.rand(K, TOPK).astype(np.float16)

    def mwrite(job):
        q = job
        my_store("matches", 256).write_rows([f"query-r{r}_{q:07d}.jpg_db-{j:05d}.jpg" for j in range(K)], {"matches0": m_blk, "matching_scores0": s_blk})
        with lock:
            counts["pairs"] += K

    def match_half():
        wp = WriterPool(mwrite, workers=1, maxsize=2, name="soak-match-writer")
        q = 0
        try:
            while time.perf_counter() < stop_at[0]:
                wp.put(q)
                q += 1
        finally:
            wp.close()

    # descriptor-set reads (float64 [128, 4096] = 4 MB each) from a small feature store written before the clock starts
    rstore_path = os.path.join(scratch, "dbfeats.h5")
    with fio.open_store(rstore_path, "w") as st:
        for j in range(16):
            st.write_group(f"db/{j:05d}.jpg", {"descriptors": t_de.astype(np.float64).transpose().copy()})
    rstore = fio.open_store(rstore_path, "r")

    def reader(tid):
        dst = np.empty((128, TOPK), dtype=np.float64)
        j = tid
        period = args.readers / max(args.read_rate, 1e-9)          # a real run reads a database set ONCE (it then lives in HBM): a bounded rate, not a spin
        nxt = time.perf_counter()
        while time.perf_counter() < stop_at[0]:
            np.copyto(dst, rstore[f"db/{j % 16:05d}.jpg"]["descriptors"].__array__())
            j += 1
            with lock:
                counts["sets_read"] += 1
            nxt += period
            pause = nxt - time.perf_counter()
            if pause > 0:
                time.sleep(min(pause, max(0.0, stop_at[0] - time.perf_counter())))

    # start together with the other ranks: the parent hands every child the same wall-clock start
    delay = args.start_at - time.time()
    if delay > 0:
        time.sleep(delay)
    t0 = time.perf_counter()
    cpu0 = time.process_time()              # CPU time of every thread of this process: what the rank COSTS, whatever share of the host it was given
    stop_at[0] = t0 + args.seconds
    threads = []
    if args.half in ("both", "extract", "decode"):
        threads.append(threading.Thread(target=extract_half))
    if args.half in ("both", "match"):
        threads.append(threading.Thread(target=match_half))
        threads += [threading.Thread(target=reader, args=(t,)) for t in range(args.readers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    cpu = time.process_time() - cpu0
    for st in all_stores:
        st.close()
    print(json.dumps({"rank": r, "seconds": round(dt, 3), "images_per_s": round(counts["images"] / dt, 1), "pairs_per_s": round(counts["pairs"] / dt, 1),
                      "descriptor_sets_read_per_s": round(counts["sets_read"] / dt, 1), "cpu_seconds": round(cpu, 3), "images": counts["images"],
                      "pairs": counts["pairs"], "placement": placement}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--writers", type=int, default=2)
    ap.add_argument("--readers", type=int, default=2)
    ap.add_argument("--k", type=int, default=50)
    ap.add_argument("--read-rate", type=float, default=100.0, help="descriptor sets (4 MB of float64 each) a rank reads per second from its feature store")
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--store", default="pack", choices=["pack", "h5", "auto"])
    ap.add_argument("--no-affinity", action="store_true")
    ap.add_argument("--half", default="both", choices=["both", "extract", "match", "decode"])
    ap.add_argument("--host-cast", action="store_true", help="the writers cast and transpose the descriptors on the host (the drivers until round 5)")
    ap.add_argument("--breakdown", action="store_true", help="besides the full legs: decode only, extract half only, match half only (pinned), and one rank alone")
    ap.add_argument("--files", type=int, default=96, help="distinct JPEG files (1600x1200, quality 90) the ranks cycle over")
    ap.add_argument("--per-rank-device-rate", type=float, default=715.0, help="images/s one GPU extracts + matches (bench.py's value)")
    ap.add_argument("--scratch", default=None)
    ap.add_argument("--images", default=None)
    ap.add_argument("--child", type=int, default=None)
    ap.add_argument("--start-at", type=float, default=0.0)
    args = ap.parse_args()
    if args.child is not None:
        return child(args)
    import importlib.util
    spec = importlib.util.spec_from_file_location("pipeline_bench", os.path.join(ROOT, "tools", "pipeline_bench.py"))
    pb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pb)
    scratch = args.scratch or tempfile.mkdtemp(prefix="sfd2_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") and args.scratch is None else None)
    images = os.path.join(scratch, "images")
    names, mean_bytes = pb.write_images(images, args.files, 0, 1200, 1600)
    legs = {}

    def leg(label, half, ranks, extra):
        start_at = time.time() + 6.0 + 0.5 * ranks          # (interpreter + numpy import of every child)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(r), "--ranks", str(args.ranks), "--workers", str(args.workers),
                                   "--writers", str(args.writers), "--readers", str(args.readers), "--k", str(args.k), "--read-rate", str(args.read_rate),
                                   "--seconds", str(args.seconds), "--store", args.store, "--half", half, "--scratch", os.path.join(scratch, label),
                                   "--images", images, "--start-at", repr(start_at)] + (["--host-cast"] if args.host_cast else []) + extra, stdout=subprocess.PIPE, text=True) for r in range(ranks)]
        per = []
        for p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise SystemExit(f"child failed ({p.returncode})")
            per.append(json.loads(out.strip().splitlines()[-1]))
        import shutil
        shutil.rmtree(os.path.join(scratch, label), ignore_errors=True)
        legs[label] = {"half": half, "ranks_running": ranks, "images_per_s": round(sum(c["images_per_s"] for c in per), 1),
                       "pairs_per_s": round(sum(c["pairs_per_s"] for c in per), 1),
                       "descriptor_sets_read_per_s": round(sum(c["descriptor_sets_read_per_s"] for c in per), 1),
                       "per_rank_images_per_s": [c["images_per_s"] for c in per], "per_rank_pairs_per_s": [c["pairs_per_s"] for c in per],
                       "placement": per[0]["placement"]}
        cpu, ni, npairs = sum(c["cpu_seconds"] for c in per), sum(c["images"] for c in per), sum(c["pairs"] for c in per)
        legs[label]["cpus_busy"] = round(cpu / max(per[0]["seconds"], 1e-9), 1)
        if half in ("decode", "extract") and ni:
            legs[label]["cpu_ms_per_image"] = round(cpu * 1e3 / ni, 2)
        if half == "match" and npairs:
            legs[label]["cpu_us_per_pair"] = round(cpu * 1e6 / npairs, 2)

    if not args.no_affinity:
        leg("pinned", args.half, args.ranks, [])
    leg("floating", args.half, args.ranks, ["--no-affinity"])
    if args.breakdown:
        pin = [] if not args.no_affinity else ["--no-affinity"]
        leg("decode_only", "decode", args.ranks, pin)
        leg("extract_half_only", "extract", args.ranks, pin)
        leg("match_half_only", "match", args.ranks, pin)
        leg("one_rank_alone_decode_only", "decode", 1, pin)
        leg("one_rank_alone", args.half, 1, pin)
    import shutil
    shutil.rmtree(scratch, ignore_errors=True)
    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    cpu_max = read("/sys/fs/cgroup/cpu.max")
    quota = None
    if cpu_max and cpu_max.split()[0] != "max":
        quota = float(cpu_max.split()[0]) / float(cpu_max.split()[1])
    need_img = args.ranks * args.per_rank_device_rate
    need_pairs = need_img * args.k
    best = legs.get("pinned") or legs["floating"]
    res = {"what": "host half of an N-rank run on THIS host, device stage stubbed: decode pool + float64 feature groups + match-store appends + descriptor reads, all ranks at once",
           "ranks": args.ranks, "decoder_threads_per_rank": args.workers, "writer_threads_per_rank": args.writers, "reader_threads_per_rank": args.readers,
           "store": args.store, "seconds": args.seconds, "jpeg_mean_bytes": round(mean_bytes), "logical_cpus": os.cpu_count(), "numa_nodes": len(node_cpulists()),
           "needed": {"images_per_s": need_img, "pairs_per_s": need_pairs, "from": f"{args.ranks} x {args.per_rank_device_rate} images/s per GPU x K = {args.k}"},
           "legs": legs, "scratch": ("tmpfs (/dev/shm): the disk is not part of this measurement" if scratch.startswith("/dev/shm") else scratch),
           "container_cpu_quota": {"cgroup_cpu_max": cpu_max, "cpus": quota,
                                   "note": ("every aggregate above is capped by this quota, not by the host: read the per-unit CPU costs" if quota else "no CPU quota on this container")},
           "cpu_budget": None,
           "verdict": {"images": "host sustains the device rate" if best["images_per_s"] >= need_img else "HOST-BOUND: decode + feature-store side is the bottleneck",
                       "pairs": "host sustains the device rate" if best["pairs_per_s"] >= need_pairs else "HOST-BOUND: match-store side is the bottleneck",
                       "images_ratio": round(best["images_per_s"] / need_img, 3), "pairs_ratio": round(best["pairs_per_s"] / need_pairs, 3)}}
    ex, ma = legs.get("extract_half_only"), legs.get("match_half_only")
    if ex and ma and "cpu_ms_per_image" in ex and "cpu_us_per_pair" in ma:
        cpus_needed = need_img * ex["cpu_ms_per_image"] * 1e-3 + need_pairs * ma["cpu_us_per_pair"] * 1e-6
        res["cpu_budget"] = {"cpu_ms_per_image_decode_to_stored_group": ex["cpu_ms_per_image"], "cpu_ms_per_image_decode_only": legs.get("decode_only", {}).get("cpu_ms_per_image"),
                             "cpu_us_per_stored_pair": ma["cpu_us_per_pair"], "cpus_needed_for_all_ranks_at_device_rate": round(cpus_needed, 1),
                             "host_logical_cpus": os.cpu_count(), "container_quota_cpus": quota,
                             "reading": (f"{args.ranks} ranks at {args.per_rank_device_rate} images/s each cost {cpus_needed:.0f} CPUs of host work "
                                         f"({os.cpu_count()} logical CPUs on this host" + (f"; this container may use {quota:.0f}" if quota else "") + ")")}
        if quota and cpus_needed > quota:
            res["verdict"]["named_bottleneck"] = (f"the container's CPU quota ({quota:.0f} CPUs, cgroup cpu.max {cpu_max}): {args.ranks} ranks need {cpus_needed:.0f}; "
                                                  f"one rank needs {cpus_needed / args.ranks:.1f}")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
