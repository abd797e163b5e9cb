#!/usr/bin/env python3
"""Per-layer device times of one 1600x1200 top-4096 extract in a given precision mode (default f32)."""
import ctypes
import sys

import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval()
m.load_state_dict(synth.make_state_dict(0))
m.cuda(0)
ctx = m.context
H, W, K = 1200, 1600, 4096
img = torch.from_numpy(synth.make_image(H, W, 100)).cuda()
kp = torch.empty((K, 2), device="cuda"); sc = torch.empty((K,), device="cuda"); de = torch.empty((K, 128), device="cuda"); n = ctypes.c_int()


def run(reps):
    for _ in range(reps):
        _lib.check(ctx.lib.sfd2_extract(ctx.h, img.data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(),
                                        de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()


run(3)
ctx.set_profiling(8)
run(3)
tot = 0.0
for r in ctx.layer_timings():
    ms = r["ms_total"] / max(1, r["launches"])
    tot += ms
    print(f"{r['name']:16s} {r['kernel']:30s} {ms * 1e3:9.1f} us  {r['flops'] / 1e9:8.2f} GF  {r['flops'] / (ms * 1e-3) / 1e12 if ms > 0 else 0:7.1f} TF/s")
print(f"total {tot:.3f} ms")
