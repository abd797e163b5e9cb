#!/usr/bin/env python3
"""Developer tool (GPU box): descriptor error of f16c against the fp32 oracle -- maximum AND root mean square over the common key points,
plus the backbone output's relative error -- for one library / option set over the heavy-tailed weight family (Student-t filters) and the
default one.  The maximum over 5e5 descriptor elements moves by 20 % from one rounding pattern to the next; the rms is what separates two
formats of the correction records.
    python tools/fp6_compare.py [--lib build/variants/libX.so] [--opts fp6_acts=0] [--cache /tmp/fp6_oracle]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--opts", default="")
ap.add_argument("--cache", default="/tmp/fp6_oracle")
ap.add_argument("--big", type=int, default=1)
args = ap.parse_args()
from sfd2_amd import _lib  # noqa: E402
if args.lib:
    _lib.use_library(args.lib)
import torch  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from sfd2_amd import synth  # noqa: E402
from sfd2_amd.extractor import extract_resnet_return  # noqa: E402
from sfd2_amd.model import ResSegNetV2  # noqa: E402

os.makedirs(args.cache, exist_ok=True)
cases = [("student", s, 480, 640, 1024) for s in range(5)] + [("default", 0, 480, 640, 1024), ("calibrated", 2, 480, 640, 1024)]
if args.big:
    cases.append(("student", 3, 1200, 1600, 4096))
tot = []
for fam, seed, h, w, k in cases:
    sd = synth.make_state_dict(seed) if fam == "default" else synth.make_state_dict(seed, family=fam)
    img = synth.make_image(h, w, 40 + seed)
    f = os.path.join(args.cache, f"{fam}_{seed}_{h}x{w}.npz")
    if os.path.exists(f):
        z = np.load(f)
        want = {"keypoints": z["kp"], "descriptors": z["de"], "trunk": z["trunk"] if "trunk" in z else None}
    else:
        want = orc.extract_resnet_return(sd, img, conf_th=0.001, topK=k)
        trunk = None
        if h <= 480:
            taps = {}
            orc.det(sd, orc.norm_rgb(img), taps)
            trunk = taps["conv4.2"]
        np.savez(f, kp=want["keypoints"], de=want["descriptors"], **({"trunk": trunk} if trunk is not None else {}))
        want = {"keypoints": want["keypoints"], "descriptors": want["descriptors"], "trunk": trunk}
    m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
    m.load_state_dict(sd)
    m.cuda(0)
    for kv in filter(None, args.opts.split("+")):
        m.context.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    got = extract_resnet_return(m, torch.from_numpy(img)[None].cuda(), conf_th=0.001, topK=k, scales=[1.0])
    a = {(float(x), float(y)): i for i, (x, y) in enumerate(got["keypoints"])}
    b = {(float(x), float(y)): i for i, (x, y) in enumerate(want["keypoints"])}
    common = sorted(set(a) & set(b))
    d = got["descriptors"][[a[c] for c in common]] - np.asarray(want["descriptors"], dtype=np.float64)[[b[c] for c in common]]
    line = f"{fam:10s} seed {seed} {w}x{h}: desc max {np.abs(d).max():.2e} rms {np.sqrt((d * d).mean()):.3e}"
    if want["trunk"] is not None:
        m.det(orc.norm_rgb(img)[None])
        t = m.context.debug_activation("conv4.2")
        e = t - want["trunk"]
        line += f"   trunk max {np.abs(e).max() / np.abs(want['trunk']).max():.2e} rms {np.sqrt((e * e).mean()) / np.sqrt((want['trunk'] ** 2).mean()):.3e} (of max / of rms)"
        tot.append(np.sqrt((e * e).mean()) / np.sqrt((want['trunk'] ** 2).mean()))
    print(line, flush=True)
print(f"[{args.lib or 'default lib'} {args.opts}] mean trunk rms error {np.mean(tot):.3e}")
