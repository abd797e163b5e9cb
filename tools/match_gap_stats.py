#!/usr/bin/env python3
"""How often does the matcher's arithmetic decide a match differently from fp64?  

SURVEY 8c's contract: matches identical wherever the top-1 / top-2 similarity gap exceeds 1e-4; scores within 1e-3.
SFD2_SIM_F16 (fp16 operands, what bench.py runs) is tested at gap > 1e-3, SFD2_SIM_F16X2 at 1e-5.  This tool measures, on
descriptor sets the NETWORK produces (two views of one scene: an image and a shifted, re-noised copy, so that true
correspondences and near-duplicates both exist), per direction of the mutual check:
  * the share of rows whose fp64 top-1 / top-2 gap lies below 1e-4, in (1e-4, 1e-3], above 1e-3;
  * per band, the share of rows where the mode's arg-max differs from fp64's;
  * the share of MATCHES (after the mutual check) that differ from the fp64 matcher, and the score error;
  * the time of one K = 50 launch in both modes.
    python tools/match_gap_stats.py [--size 1600x1200 --topk 4096 --views 6]
Prints one JSON line (committed under profiles/).  The fp64 matcher is numpy on the host (test infrastructure only)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1600x1200")
    ap.add_argument("--topk", type=int, default=4096)
    ap.add_argument("--views", type=int, default=6)
    args = ap.parse_args()
    W, H = (int(v) for v in args.size.split("x"))
    import torch
    from sfd2_amd import _lib, synth
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    model = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    model.load_state_dict(synth.make_state_dict(0))
    model.cuda(0)
    ctx = model.context
    rs = np.random.RandomState(7)
    base = synth.make_image(H, W, 11)
    sets = []
    for v in range(args.views):
        dy, dx = (0, 0) if v == 0 else (int(rs.randint(-24, 25)), int(rs.randint(-24, 25)))
        img = np.roll(base, (dy, dx), axis=(1, 2))
        if v:
            img = np.clip(img * (1.0 + 0.1 * rs.randn()) + 0.02 * rs.standard_normal(img.shape).astype(np.float32), 0.0, 1.0).astype(np.float32)
        sets.append(extract_resnet_return(model, img[None], conf_th=0.001, topK=args.topk)["descriptors"])
    bands = {"gap<=1e-4": (0.0, 1e-4), "1e-4<gap<=1e-3": (1e-4, 1e-3), "gap>1e-3": (1e-3, 10.0)}
    out = {"workload": f"{args.views} views of one synthetic scene {W}x{H}, top-{args.topk} f16c descriptors; view 0 against each other view, both directions",
           "rows": 0, "bands": {k: {"rows": 0, "argmax_differs_f16": 0, "argmax_differs_f16x2": 0} for k in bands},
           "matches_fp64": 0, "matches_differ_f16": 0, "matches_differ_f16x2": 0, "score_err_max_f16": 0.0, "score_err_max_f16x2": 0.0}

    def device_match(d0, d1, sim_mode, mutual):
        a0 = np.ascontiguousarray(d0, dtype=np.float32)
        a1 = np.ascontiguousarray(d1, dtype=np.float32)
        conf = _lib.MatchConf(_lib.MATCH_HLOC, mutual, 0.0, 0.0, sim_mode)
        m = np.empty((len(a0),), dtype=np.int64)
        s = np.empty((len(a0),), dtype=np.float32)
        _lib.check(ctx.lib.sfd2_match(ctx.h, a0.ctypes.data, len(a0), a1.ctypes.data, len(a1), 128, _lib.DT_F32, _lib.LAYOUT_ND, 0,
                                      ctypes.byref(conf), m.ctypes.data, s.ctypes.data, 0))
        return m, s

    q = sets[0]
    for v in range(1, args.views):
        for d0, d1 in ((q, sets[v]), (sets[v], q)):
            sim = d0.astype(np.float32).astype(np.float64) @ d1.astype(np.float32).astype(np.float64).T     # the reference's operands are fp32 (.float())
            order = np.argsort(-sim, axis=1)[:, :2]
            top1 = sim[np.arange(len(sim)), order[:, 0]]
            gap = top1 - sim[np.arange(len(sim)), order[:, 1]]
            got = {name: device_match(d0, d1, mode, 0) for name, mode in (("f16", _lib.SIM_F16), ("f16x2", _lib.SIM_F16X2))}   # ONN: arg-max of every row
            out["rows"] += len(sim)
            for k, (lo, hi) in bands.items():
                sel = (gap > lo) & (gap <= hi) if lo > 0 else (gap <= hi)
                out["bands"][k]["rows"] += int(sel.sum())
                for name in ("f16", "f16x2"):
                    out["bands"][k][f"argmax_differs_{name}"] += int((got[name][0][sel] != order[sel, 0]).sum())
        # the mutual matcher proper, query -> view
        sim = q.astype(np.float32).astype(np.float64) @ sets[v].astype(np.float32).astype(np.float64).T
        nn01, nn10 = sim.argmax(1), sim.argmax(0)
        want = np.where(nn10[nn01] == np.arange(len(sim)), nn01, -1)
        out["matches_fp64"] += int((want >= 0).sum())
        for name, mode in (("f16", _lib.SIM_F16), ("f16x2", _lib.SIM_F16X2)):
            m, s = device_match(q, sets[v], mode, 1)
            out[f"matches_differ_{name}"] += int((m != want).sum())
            sc = (sim.max(1) + 1.0) / 2.0
            hit = (m >= 0) & (want >= 0)
            out[f"score_err_max_{name}"] = max(out[f"score_err_max_{name}"], float(np.abs(s[hit] - sc[hit]).max()))
    for k in bands:
        b = out["bands"][k]
        b["share_of_rows"] = round(b["rows"] / max(1, out["rows"]), 6)
    # time of one K = 50 launch, device-resident fp16 sets, both modes
    g = torch.Generator(device="cpu").manual_seed(1)
    def unit(n):
        d = torch.randn(n, 128, generator=g)
        return d / d.norm(dim=1, keepdim=True)
    K, N = 50, 4096
    qd = unit(N).cuda()
    db = [unit(N).cuda().contiguous() for _ in range(K)]
    dsets = (_lib.DescSet * K)(*[_lib.DescSet(d.data_ptr(), N, _lib.DT_F32, _lib.LAYOUT_ND, 1) for d in db])
    qs = _lib.DescSet(qd.data_ptr(), N, _lib.DT_F32, _lib.LAYOUT_ND, 1)
    mo = torch.empty((K, N), dtype=torch.int64, device="cuda")
    so = torch.empty((K, N), dtype=torch.float32, device="cuda")
    for name, mode in (("f16", _lib.SIM_F16), ("f16x2", _lib.SIM_F16X2)):
        conf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, mode)
        def call():
            _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qs), dsets, K, 128, ctypes.byref(conf), mo.data_ptr(), so.data_ptr(), 1, _lib.FLAG_ASYNC))
        for _ in range(5):
            call()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        ctx.sync()
        out[f"ms_per_k50_launch_{name}"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
