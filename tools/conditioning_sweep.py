#!/usr/bin/env python3
"""Where does the compensated mode (f16c) stop holding north_star's 1e-3 on descriptors?  CPU study on the torch twin
(tools/error_budget_fp6.py's CompTwin = the f16c arithmetic of DESIGN section 3 with e4m3 correction operands, plus the
+-1792 saturation of the compensated tensors), over the weight families of sfd2_amd.synth.make_state_dict and
power-of-two gains on the stored tensors.  Errors are against the all-fp32 twin on the same weights.
    python tools/conditioning_sweep.py [HxW]        (default 240x320)
The GPU-side counterpart that asserts the envelope is tests/test_gpu_f16c_conditioning.py."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import torch_twin as tt      # noqa: E402
from sfd2_amd import synth               # noqa: E402
import error_budget_fp6 as eb            # noqa: E402

SAT = 1792.0


class CompTwinSat(eb.CompTwin):
    """CompTwin + the saturation of compensated tensors and a record of every stored tensor's maximum."""

    def __init__(self, sd):
        super().__init__(sd, "e4m3")
        self.amax = {}
        self.nsat = {}

    def layer(self, name, x, conv, bn, stride=1, relu=True, groups=1, residual=None):
        y = super().layer(name, x, conv, bn, stride, relu, groups, residual)
        head = name.startswith(("convP", "convD", "ConvSta"))
        self.amax[name] = float(y.abs().max())
        if not head and name != "conv1a.never":
            inner = name.startswith("conv4.") and (name.endswith("conv1") or name.endswith("conv2"))
            lim = 65504.0 if inner else SAT
            self.nsat[name] = int((y.abs() > lim).sum())
            y = y.clamp(-lim, lim)
        return y


def f16_policy():
    pol = {}
    heads = ("convPb", "convDb", "ConvSta")
    names = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b"] + [f"conv4.{b}.conv{i}" for b in range(3) for i in (1, 2, 3)] + \
            ["convPa.0", "convPa.3", "convPb", "convDa.0", "convDa.3", "convDb", "ConvSta"]
    for n in names:
        pol[n] = tt.Policy(w="f16", x="f16" if n == "conv1a" else "f32", out="f32" if n in heads else "f16")
    pol["ConvSta"] = tt.Policy()
    return pol


def one(sd, img, label):
    H, W = img.shape[1:]
    x = tt.norm_rgb(torch.from_numpy(img)[None])
    with torch.no_grad():
        ref_tw = tt.Twin(sd)
        ref_tw.taps = {}
        logits, draw, _ = ref_tw.det_raw(x)
        ref = F.normalize(draw, dim=1)
        ex = tt.extract(ref_tw, img, topK=1024)
        kp = ex["keypoints"]
        if len(kp) == 0:
            print(f"{label:44s} (no key points in the fp32 run)")
            return
        ref_s = eb.sample(ref, kp, H, W)
        bb = [k for k in ref_tw.taps if not k.startswith(("convP", "convD", "ConvSta"))]
        amax = max(float(ref_tw.taps[k].abs().max()) for k in bb)
        amin = min(float(ref_tw.taps[k].abs().max()) for k in bb)
        ct = CompTwinSat(sd)
        lg, d, _ = ct.det_raw(x)
        d = F.normalize(d, dim=1)
        e_c = float((eb.sample(d, kp, H, W) - ref_s).abs().max())
        nsat = sum(ct.nsat.values())
        t16 = tt.Twin(sd, f16_policy())
        _, d16, _ = t16.det_raw(x)
        d16 = F.normalize(d16, dim=1)
        e_16 = float((eb.sample(d16, kp, H, W) - ref_s).abs().max())
        exc = tt.extract(ct, img, topK=1024)
        a = {tuple(p) for p in exc["keypoints"].astype(int)}
        b = {tuple(p) for p in kp.astype(int)}
        iou = len(a & b) / max(1, len(a | b))
    print(f"{label:44s} {amin:9.3g} {amax:9.3g} {nsat:7d} {e_c:10.2e} {e_16:10.2e} {iou:7.3f}  {len(kp)}", flush=True)


def main():
    H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "240x320").split("x"))
    img = synth.make_image(H, W, 5)
    print(f"# {H}x{W}; sampled-descriptor max error against the fp32 twin on the same weights")
    print(f"{'weights':44s} {'min max|x|':>9s} {'max max|x|':>9s} {'n sat':>7s} {'f16c':>10s} {'f16':>10s} {'kp IoU':>7s}  n_kp")
    for fam in (None, "student", "calibrated", "biased", "dead", "smallvar"):
        for seed in (0, 1, 2):
            one(synth.make_state_dict(seed, family=fam), img, f"{fam or 'gauss'} seed {seed}")
    for k in (-10, -8, -6, -4, -2, 2, 4, 6, 8, 10, 12):
        one(synth.make_state_dict(0, gain_log2=k), img, f"gauss seed 0, all tensors x 2^{k}")
    for on in ("conv1a", "conv2a", "conv3a", "trunk", "t1", "t2"):
        for k in (-8, 8):
            one(synth.make_state_dict(0, gain_log2=k, gain_on=on), img, f"gauss seed 0, {on} x 2^{k}")


if __name__ == "__main__":
    main()
