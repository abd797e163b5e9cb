#!/usr/bin/env python3
"""Bit-compare the extraction outputs (key points, scores, descriptors) of two libsfd2hip builds on the same inputs -- the check for a
kernel variant that claims the same operations in the same order -- and screen the second for run-to-run differences.
    python tools/compare_libs_extract.py default build/variants/libX.so [--precision f16c] [--runs 6]
(a library may carry environment switches for its worker: default@SFD2_AB_OPTS=s2d=0)"""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SIZES = [(1200, 1600), (1063, 1600), (480, 640), (133, 211), (96, 128), (1600, 1200)]
WORKER = r'''
import sys, os, numpy as np
sys.path.insert(0, %r)
from sfd2_amd import _lib
if sys.argv[1] != "default":
    _lib.use_library(sys.argv[1])
from sfd2_amd import synth
from sfd2_amd.model import ResSegNetV2
from sfd2_amd.extractor import extract_resnet_return
m = ResSegNetV2(outdim=128, require_stability=True, precision=sys.argv[3]).eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
for kv in filter(None, os.environ.get("SFD2_AB_OPTS", "").split("+")):
    m.context.set_option(kv.split("=")[0], int(kv.split("=")[1]))
out = {}
for (h, w) in %r:
    x = synth.make_image(h, w, 7 + h).astype(np.float32)
    first = None
    for r in range(int(sys.argv[4])):
        g = extract_resnet_return(m, x[None], conf_th=0.001, topK=4096, scales=[1.0])
        if first is None:
            first = g
        else:
            for k in ("keypoints", "scores", "descriptors"):
                if not np.array_equal(g[k], first[k]):
                    print("NONDETERMINISTIC", sys.argv[1], (h, w), k, "run", r, flush=True)
    for k in ("keypoints", "scores", "descriptors"):
        out[f"{h}x{w}/{k}"] = first[k]
np.savez(sys.argv[2], **out)
''' % (os.path.abspath(ROOT), SIZES)


def split_spec(spec):
    lib, _, envs = spec.partition("@")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    return lib, env


ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs=2)
ap.add_argument("--precision", default="f16c")
ap.add_argument("--runs", type=int, default=6)
args = ap.parse_args()
res = []
with tempfile.TemporaryDirectory() as td:
    for i, spec in enumerate(args.libs):
        lib, env = split_spec(spec)
        f = os.path.join(td, f"o{i}.npz")
        r = subprocess.run([sys.executable, "-c", WORKER, lib, f, args.precision, str(args.runs)], env=env, capture_output=True, text=True)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        res.append(dict(np.load(f)))
bad = 0
for k in sorted(res[0]):
    a, b = res[0][k], res[1][k]
    same = a.shape == b.shape and np.array_equal(a, b)
    if not same:
        bad += 1
        d = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape else float("nan")
        print(f"DIFFERENT {k}: shapes {a.shape} {b.shape}, max abs diff {d:.3e}")
print(f"{args.libs[0]} vs {args.libs[1]} ({args.precision}): {len(res[0]) - bad} of {len(res[0])} arrays bit-identical")
sys.exit(1 if bad else 0)
