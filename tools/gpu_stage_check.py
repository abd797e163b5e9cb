#!/usr/bin/env python3
"""Developer tool (GPU box): stage-by-stage comparison of libsfd2hip against the CPU oracle.
Prints one line per check and never stops at the first failure.
    python tools/gpu_stage_check.py [--big]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from sfd2_amd import _lib, synth  # noqa: E402
from sfd2_amd.extractor import extract_resnet_return  # noqa: E402
from sfd2_amd.model import ResSegNetV2  # noqa: E402

np.set_printoptions(linewidth=200, precision=5, suppress=True)
OK = True


def report(name, ok, msg):
    global OK
    OK = OK and ok
    print(f"[{'ok ' if ok else 'BAD'}] {name:34s} {msg}", flush=True)


def cmp(name, got, want, atol, rtol=0.0):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        report(name, False, f"shape {got.shape} vs {want.shape}")
        return
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    scale = np.abs(want).max() if want.size else 0
    report(name, not bad.any(), f"max|err| {err.max() if err.size else 0:.3e}  ref max {scale:.3e}  bad {bad.mean() if bad.size else 0:.2%}")
    if bad.any() and got.ndim == 3:
        c, y, x = np.unravel_index(np.argmax(err), err.shape)
        print(f"        worst at c={c} y={y} x={x}: got {got[c,y,x]:.5f} want {want[c,y,x]:.5f}; "
              f"per-channel bad {bad.reshape(bad.shape[0],-1).mean(1)[:8]} rows bad {bad.mean(axis=(0,2))[:8]}")


def check_det(model, sd, h, w, seed):
    img = synth.make_image(h, w, seed)
    x = orc.norm_rgb(img)
    taps = {}
    t0 = time.time()
    o_score, o_stab, o_desc = orc.det(sd, x, taps)
    print(f"-- det {h}x{w}: oracle {time.time()-t0:.1f}s")
    score, stab, desc = model.det(x[None])
    ctx = model.context
    order = ["conv1a", "bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "conv4.0.bn1", "conv4.0.bn2", "conv4.0", "conv4.1",
             "conv4.2", "convPa", "convDa", "convPb", "convDb", "ConvSta"]
    for name in order:
        got = ctx.debug_activation(name)
        want = taps[name]
        cmp(f"act {name}", got, want, atol=2e-2 * max(1.0, float(np.abs(want).max()) / 4), rtol=2e-2)
    cmp("det score", score[0, 0], o_score, atol=2e-3)
    cmp("det desc", desc[0], o_desc, atol=2e-3)
    mism = (stab[0, 0] != o_stab).mean()
    report("det stability", mism < 0.02, f"mismatch {mism:.3%}")
    return taps, o_score, o_stab, o_desc


def main():
    big = "--big" in sys.argv
    sd = synth.make_state_dict(0)
    model = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
    model.load_state_dict(sd)
    model.cuda()
    ctx = model.context
    lib = ctx.lib

    taps, o_score, o_stab, o_desc = check_det(model, sd, 64, 96, 11)
    taps2, o_score2, o_stab2, o_desc2 = check_det(model, sd, 100, 130, 12)

    # ---- stage: heat map (bit exact given the oracle's score + ConvSta logits)
    for (h, w, sc, tp) in ((64, 96, o_score, taps), (100, 130, o_score2, taps2)):
        sta = np.ascontiguousarray(tp["ConvSta"], dtype=np.float32)
        want = orc.heatmap(sc, orc.cls_to_value(orc.resize_bilinear(sta, h, w)), h, w)
        got = np.empty((h, w), dtype=np.float32)
        sc32 = np.ascontiguousarray(sc, dtype=np.float32)
        _lib.check(lib.sfd2_heatmap(ctx.h, sc32.ctypes.data, sc.shape[0], sc.shape[1], sta.ctypes.data, sta.shape[1], sta.shape[2],
                                    h, w, got.ctypes.data))
        report(f"heatmap {h}x{w} bit-exact", np.array_equal(got, want), f"mismatch {(got != want).mean():.3%} max {np.abs(got-want).max():.2e}")

    # ---- stage: NMS / selection (bit exact)
    rs = np.random.RandomState(3)
    maps = {"rand 61x83": rs.random_sample((61, 83)), "rand 200x333": rs.random_sample((200, 333)),
            "plateau 64x64": np.floor(rs.random_sample((64, 64)) * 4) / 4,
            "sparse 70x90": np.where(rs.random_sample((70, 90)) > 0.97, rs.random_sample((70, 90)), 0),
            "tiny 8x9": rs.random_sample((8, 9)), "oracle heat": orc.heatmap(o_score2, o_stab2, 100, 130)}
    for name, m in maps.items():
        m = np.ascontiguousarray(m, dtype=np.float32)
        h, w = m.shape
        want = orc.simple_nms(m, 4)
        got = np.empty_like(m)
        _lib.check(lib.sfd2_simple_nms(ctx.h, m.ctypes.data, h, w, 4, got.ctypes.data))
        report(f"nms {name}", np.array_equal(got, want), f"mismatch {(got != want).sum()} of {m.size}")
        for topk in (50, -1):
            kp_w, sc_w, _ = orc.select_keypoints(want, 0.001, 4, topk)
            cap = h * w
            kp = np.empty((cap, 2), dtype=np.float32)
            sc = np.empty((cap,), dtype=np.float32)
            n = ctypes.c_int()
            _lib.check(lib.sfd2_select_keypoints(ctx.h, m.ctypes.data, h, w, 0.001, 4, 4, topk, kp.ctypes.data, sc.ctypes.data, cap, ctypes.byref(n)))
            ok = n.value == len(sc_w) and np.array_equal(kp[:n.value], kp_w) and np.array_equal(sc[:n.value], sc_w)
            report(f"select {name} k={topk}", ok, f"n {n.value} vs {len(sc_w)}")

    # ---- stage: descriptor sampling
    kp_w, sc_w, _ = orc.select_keypoints(orc.simple_nms(orc.heatmap(o_score2, o_stab2, 100, 130), 4), 0.001, 4, -1)
    want = orc.sample_descriptors(o_desc2, kp_w, 100, 130)
    got = np.empty_like(want)
    dm = np.ascontiguousarray(o_desc2, dtype=np.float32)
    _lib.check(lib.sfd2_sample_descriptors(ctx.h, dm.ctypes.data, dm.shape[1], dm.shape[2], 100, 130, kp_w.ctypes.data, len(kp_w), got.ctypes.data))
    cmp("sample_desc 100x130", got, want, atol=2e-6)

    # ---- end to end extract
    for (h, w, seed, topk) in ((96, 128, 21, 200), (100, 130, 22, -1), (480, 640, 0, 1024)):
        img = synth.make_image(h, w, seed)
        t0 = time.time()
        want = orc.extract_resnet_return(sd, img, 0.001, topk)
        t1 = time.time()
        got = extract_resnet_return(model, img[None], conf_th=0.001, topK=topk, scales=[1.0])
        a = {(int(x), int(y)): i for i, (x, y) in enumerate(got["keypoints"])}
        b = {(int(x), int(y)): i for i, (x, y) in enumerate(want["keypoints"])}
        common = sorted(set(a) & set(b))
        iou = len(common) / max(1, len(set(a) | set(b)))
        ia = np.array([a[k] for k in common], dtype=int)
        ib = np.array([b[k] for k in common], dtype=int)
        ds = np.abs(got["scores"][ia] - want["scores"][ib]).max() if common else 0
        dd = np.abs(got["descriptors"][ia] - want["descriptors"][ib]).max() if common else 0
        report(f"extract {h}x{w} k={topk}", iou > 0.9 and dd < 5e-3,
               f"n {len(a)} vs {len(b)}  IoU {iou:.4f}  max|dscore| {ds:.2e}  max|ddesc| {dd:.2e}  (oracle {t1-t0:.1f}s) tim {ctx.timings()}")

    # ---- matchers
    for (n0, n1) in ((1024, 777), (300, 512), (4096, 4096)):
        g0 = synth.make_descriptors(n0, seed=n0)
        g1 = synth.make_descriptors(n1, seed=n1 + 1)
        k = min(n0, n1) // 2
        rs = np.random.RandomState(9)
        src, dst = rs.permutation(n0)[:k], rs.permutation(n1)[:k]
        noisy = g0[src] + (0.02 + 0.1 * rs.random_sample((k, 1))).astype(np.float32) * rs.standard_normal((k, 128)).astype(np.float32)
        g1[dst] = noisy / np.linalg.norm(noisy, axis=1, keepdims=True)
        confs = {"NNM": dict(do_mutual_check=True), "ONN": dict(do_mutual_check=False),
                 "NNR": dict(do_mutual_check=True, distance_threshold=0.9), "RATIO": dict(do_mutual_check=True, ratio_threshold=0.8)}
        from sfd2_amd.matchers.nearest_neighbor import NearestNeighbor
        from sfd2_amd.matcher import Matcher, confs as mconfs
        for sim_mode in ("f16", "f16x2"):
            for name, cf in confs.items():
                want = orc.hloc_nearest_neighbor(g0, g1, **cf)
                got = NearestNeighbor({**cf, "sim_mode": sim_mode})({"descriptors0": g0.T[None].copy(), "descriptors1": g1.T[None].copy()})
                diff = (got["matches0"][0] != want["matches0"]).mean()
                same = got["matches0"][0] == want["matches0"]
                ds = np.abs(got["matching_scores0"][0][same] - want["matching_scores0"][same]).max()
                report(f"hloc {name} {n0}x{n1} {sim_mode}", diff < (0.02 if sim_mode == "f16" else 0.002) and ds < 1e-3, f"match diff {diff:.3%} max|dscore| {ds:.2e}")
            for name in ("NNM", "NNR"):
                mc = {"output": name, "model": {**mconfs[name]["model"], "sim_mode": sim_mode}}
                want = orc.itloc_matcher(g0, g1, mc["model"]["name"], 0.9)
                got = Matcher(mc)({"descriptors0": g0.astype(np.float64), "descriptors1": g1.astype(np.float64)})
                diff = (got["matches0"] != want["matches0"]).mean()
                ds = np.abs(got["matching_scores0"] - want["matching_scores0"]).max()
                report(f"itloc {name} {n0}x{n1} {sim_mode}", diff < (0.02 if sim_mode == "f16" else 0.002) and ds < 1e-3, f"match diff {diff:.3%} max|dscore| {ds:.2e}")
        print("   match timing:", _lib.default_context(0).timings())

    if big:
        img = synth.make_image(1200, 1600, 5)
        for i in range(3):
            got = extract_resnet_return(model, img[None], conf_th=0.001, topK=4096, scales=[1.0])
            print("1200x1600:", len(got["scores"]), ctx.timings())
    print("ALL OK" if OK else "SOME CHECKS FAILED")


if __name__ == "__main__":
    main()
