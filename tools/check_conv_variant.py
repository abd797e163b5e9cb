#!/usr/bin/env python3
"""Compare the 3x3-layer activations of two libsfd2hip builds (layer-wise path, same inputs) and screen the second
one for run-to-run differences (hand-synchronised kernels: a race shows up as a tile that changes between runs).
    python tools/check_conv_variant.py default build/variants/libsfd2hip_pp.so
(a library may carry environment switches for its worker: default@SFD2_CONV_RF=off)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
LAYERS = ["bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "convPa.0", "convPa", "convDa.0", "convDa"]
SIZES = [(1200, 1600), (240, 320), (133, 211), (512, 384), (64, 1056)]
WORKER = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from sfd2_amd import _lib
if sys.argv[1] != "default":
    _lib.use_library(sys.argv[1])
from sfd2_amd import synth
from sfd2_amd.model import ResSegNetV2
m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
m.context.set_option("fuse", 0)
out = {}
for (h, w) in %r:
    x = synth.make_image(h, w, 7 + h).astype(np.float32)
    runs = 12 if h == 1200 else 4
    first = None
    for r in range(runs):
        m.det(x[None])
        acts = {k: m.context.debug_activation(k) for k in %r}
        if first is None:
            first = acts
        else:
            for k in acts:
                if not np.array_equal(acts[k], first[k]):
                    d = np.argwhere(acts[k] != first[k])
                    print("NONDETERMINISTIC", sys.argv[1], (h, w), k, "run", r, len(d), "elements, first at", d[0], flush=True)
    for k, v in first.items():
        out[f"{h}x{w}/{k}"] = v.astype(np.float16)
np.savez(sys.argv[2], **out)
''' % (os.path.abspath(ROOT), SIZES, LAYERS)


def split_spec(spec):
    """'lib.so@KEY=VAL,KEY2=VAL2' -> (lib path, environment of its worker)"""
    lib, _, envs = spec.partition("@")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    return lib, env

a, b = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
files = []
for i, lib in enumerate((a, b)):
    f = os.path.join(tmp, f"acts_{i}.npz")
    lib_path, env = split_spec(lib)
    subprocess.run([sys.executable, "-c", WORKER, lib_path, f], check=True, env=env)
    files.append(np.load(f))
for key in files[0].files:
    x, y = files[0][key].astype(np.float32), files[1][key].astype(np.float32)
    d = np.abs(x - y)
    rel = d.max() / max(1e-9, np.abs(x).max())
    nbad = int((d > 4e-3 * np.abs(x).max()).sum())
    print(f"{key:24s} shape {x.shape} max|a| {np.abs(x).max():8.3f} max diff {d.max():.5f} (rel {rel:.2e}) elements off by > 4e-3 max: {nbad}")
