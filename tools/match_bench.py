#!/usr/bin/env python3
"""Matcher-only timing (K x (n x n) device-resident fp16 sets, NNM) for kernel A/B runs:
    python tools/match_bench.py [--lib path/to/variant.so] [--k 50] [--n 4096] [--reps 30]
Prints one line: lib, ms per sfd2_match_batch call (HIP events through the context's timings)."""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--k", type=int, default=50)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--reps", type=int, default=30)
args = ap.parse_args()
from sfd2_amd import _lib   # noqa: E402
if args.lib:
    _lib.use_library(args.lib)
import torch   # noqa: E402

ctx = _lib.Context(0)
g = torch.Generator(device="cpu").manual_seed(1)
def unit(n):
    d = torch.randn(n, 128, generator=g)
    return d / d.norm(dim=1, keepdim=True)
q = unit(args.n).cuda()
db = [unit(args.n).to(torch.float16).cuda().contiguous() for _ in range(args.k)]
sets = (_lib.DescSet * args.k)(*[_lib.DescSet(d.data_ptr(), args.n, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in db])
qs = _lib.DescSet(q.data_ptr(), args.n, _lib.DT_F32, _lib.LAYOUT_ND, 1)
conf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)
m = torch.empty((args.k, args.n), dtype=torch.int64, device="cuda")
s = torch.empty((args.k, args.n), dtype=torch.float32, device="cuda")
def call():
    _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(qs), sets, args.k, 128, ctypes.byref(conf), m.data_ptr(), s.data_ptr(), 1, _lib.FLAG_ASYNC))
for _ in range(10):
    call()
ctx.sync()
t0 = time.perf_counter()
for _ in range(args.reps):
    call()
ctx.sync()
dt = (time.perf_counter() - t0) / args.reps
ctx.set_profiling(args.reps + 2)          # device time per kernel (HIP events around every launch)
for _ in range(args.reps):
    call()
rows = ctx.layer_timings()
ctx.set_profiling(0)
dev = ", ".join(f"{r['name']} {r['ms_total'] / max(1, r['launches']) * 1e3:.1f} us" for r in rows)
print(f"{os.path.basename(args.lib or 'libsfd2hip.so'):36s} K={args.k} n={args.n}: wall {dt * 1e3:.3f} ms/batch; device: {dev}; "
      f"matches {(m >= 0).sum().item()} checksum {int(m.sum().item())}")
