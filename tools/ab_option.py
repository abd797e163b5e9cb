#!/usr/bin/env python3
"""Interleaved A/B of a context option on the extract path (1600x1200, top-4096), one process, two contexts:
    python tools/ab_option.py sparse_desc 0 1            (SFD2_AB_PREC=f16 | f16c (default) | ...; SFD2_AB_LAYER=conv2b prints that layer's device time too)"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
H, W, K = 1200, 1600, 4096
sd = synth.make_state_dict(0)
imgs = [torch.from_numpy(synth.make_image(H, W, 100 + i)).cuda() for i in range(4)]
lanes = []
for v in vals:
    m = ResSegNetV2(outdim=128, require_stability=True, precision=os.environ.get("SFD2_AB_PREC", "f16c")).eval()
    m.load_state_dict(sd)
    m.cuda(0)
    m.context.set_option(key, v)
    lanes.append((m, torch.empty((K, 2), device="cuda"), torch.empty((K,), device="cuda"), torch.empty((K, 128), device="cuda")))
n = ctypes.c_int()


def run(l, reps):
    m, kp, sc, de = l
    ctx = m.context
    for i in range(reps):
        _lib.check(ctx.lib.sfd2_extract(ctx.h, imgs[i % 4].data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(),
                                        de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()


for l in lanes:
    run(l, 30)
res = [[] for _ in lanes]
for r in range(5):
    for i, l in enumerate(lanes):
        t0 = time.perf_counter()
        run(l, 100)
        res[i].append((time.perf_counter() - t0) / 100 * 1e3)
for v, r in zip(vals, res):
    print(f"{key}={v}: ms/extract {[round(x, 4) for x in r]}")

layer = os.environ.get("SFD2_AB_LAYER")
if layer:
    for v, l in zip(vals, lanes):
        ctx = l[0].context
        ctx.set_profiling(12)
        run(l, 10)
        rows = [r for r in ctx.layer_timings() if r["name"] == layer]
        ctx.set_profiling(0)
        for r in rows:
            print(f"{key}={v}: {layer} {r['kernel']} {r['ms_total'] / max(1, r['launches']) * 1e3:.1f} us")
