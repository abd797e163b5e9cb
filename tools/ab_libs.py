#!/usr/bin/env python3
"""Interleaved A/B of libsfd2hip builds on the extract path (1600x1200, top-4096): each library runs in its own worker
process (a process binds one .so), rounds alternate between them, per-layer device times of the last round are printed.
    python tools/ab_libs.py build/variants/libA.so build/variants/libB.so [--rounds 3] [--layers conv3b,convDa.0]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
WORKER = r'''
import sys, ctypes, time, json, os
sys.path.insert(0, %r)
from sfd2_amd import _lib
lib_path = sys.argv[1]
if lib_path != "default":
    _lib.use_library(lib_path)
import torch
from sfd2_amd import synth
from sfd2_amd.model import ResSegNetV2
m = ResSegNetV2(outdim=128, require_stability=True, precision=os.environ.get("SFD2_AB_PREC", "f16c")).eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
ctx = m.context; lib = ctx.lib
for kv in filter(None, os.environ.get("SFD2_AB_OPTS", "").split("+")):      # e.g. default@SFD2_AB_OPTS=fp6_acts=1+fp6_filters=1
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
H, W, K = 1200, 1600, 4096
imgs = [torch.from_numpy(synth.make_image(H, W, 100 + i)).cuda() for i in range(4)]
kp = torch.empty((K, 2), device="cuda"); sc = torch.empty((K,), device="cuda"); de = torch.empty((K, 128), device="cuda"); n = ctypes.c_int()
def run(reps):
    for i in range(reps):
        _lib.check(lib.sfd2_extract(ctx.h, imgs[i %% 4].data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()
run(30)
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "time":
        t0 = time.perf_counter(); run(60); print(json.dumps({"ms": (time.perf_counter() - t0) / 60 * 1e3}), flush=True)
    elif cmd == "layers":
        ctx.set_profiling(12); run(10); rows = ctx.layer_timings(); ctx.set_profiling(0)
        print(json.dumps({r["name"]: round(r["ms_total"] / max(1, r["launches"]) * 1e3, 1) for r in rows}), flush=True)
    elif cmd == "quit":
        break
''' % os.path.abspath(ROOT)


def split_spec(spec):
    """'lib.so@KEY=VAL,KEY2=VAL2' -> (lib path, environment of its worker)"""
    lib, _, envs = spec.partition("@")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    return lib, env

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
procs = [subprocess.Popen([sys.executable, "-c", WORKER, split_spec(l)[0]], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                          env=split_spec(l)[1]) for l in args.libs]
def ask(p, cmd):
    p.stdin.write(cmd + "\n"); p.stdin.flush()
    return json.loads(p.stdout.readline())
times = [[] for _ in procs]
for _ in range(args.rounds):
    for i, p in enumerate(procs):
        times[i].append(round(ask(p, "time")["ms"], 4))
layers = [ask(p, "layers") for p in procs]
for l, t in zip(args.libs, times):
    print(f"{os.path.basename(l):40s} ms/extract {t}")
names = list(layers[0].keys())
print(f"{'layer':16s} " + " ".join(f"{os.path.basename(l)[-18:]:>18s}" for l in args.libs))
for nme in names:
    print(f"{nme:16s} " + " ".join(f"{ly.get(nme, float('nan')):18.1f}" for ly in layers))
for p in procs:
    p.stdin.write("quit\n"); p.stdin.flush(); p.wait()
