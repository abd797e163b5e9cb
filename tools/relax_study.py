#!/usr/bin/env python3
"""Which compensated backbone layers could run WITHOUT their correction terms (plain fp16 K loop over the hi plane, the output's corr
plane still written from the fp32 accumulators)?  CPU study on the torch twin (tools/error_budget_fp6.py's CompTwin, fp6 records):
for every candidate set of relaxed layers the descriptor error against the all-fp32 twin -- dense map and sampled at the fp32 run's
key points -- over weight families and seeds.  VERDICT r5 item 1c.
    python tools/relax_study.py [HxW] [family ...]        (default 480x640, families default student)
"""
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


class RelaxTwin(eb.CompTwin):
    """CompTwin('fp6') with the layers in `plain` run as y = conv(fp16(x), fp16(w)) and those in `xonly` / `wonly` with one correction term."""

    def __init__(self, sd, plain=(), xonly=(), wonly=()):
        super().__init__(sd, "fp6")
        self.plain, self.xonly, self.wonly = set(plain), set(xonly), set(wonly)

    def layer(self, name, x, conv, bn, stride=1, relu=True, groups=1, residual=None):
        if name not in self.plain | self.xonly | self.wonly:
            return super().layer(name, x, conv, bn, stride, relu, groups, residual)
        w = self.sd[conv + ".weight"]
        k = w.shape[-1]
        cv = lambda a, b: F.conv2d(a, b, None, stride=stride, padding=k // 2, groups=groups)   # noqa: E731
        wh = eb.f16(w)
        xh = eb.f16(x)
        y = cv(xh, wh)
        if name in self.xonly:
            y = y + cv(eb.q_fp6_act((x - xh) * 2048.0, x) / 2048.0, eb.q_fp6_w(w, w))
        if name in self.wonly:
            y = y + cv(eb.q_fp6_act(x, x), eb.q_fp6_w((w - wh) * 2048.0, w) / 2048.0)
        a, s = self._fold(conv, bn)
        y = y * a.view(1, -1, 1, 1) + s.view(1, -1, 1, 1)
        assert residual is None
        return F.relu(y) if relu else y


def main():
    args = sys.argv[1:]
    size = (480, 640)
    if args and "x" in args[0] and args[0][0].isdigit():
        size = tuple(int(v) for v in args.pop(0).split("x"))
    fams = args or ["default", "student"]
    H, W = size
    cands = [("none (today)", {}),
             ("conv3b plain", dict(plain=["conv3b"])),
             ("conv3a plain", dict(plain=["conv3a"])),
             ("conv2b plain", dict(plain=["conv2b"])),
             ("conv2a plain", dict(plain=["conv2a"])),
             ("conv3b x-term only", dict(xonly=["conv3b"])),
             ("conv3b w-term only", dict(wonly=["conv3b"])),
             ("conv3a+conv3b plain", dict(plain=["conv3a", "conv3b"])),
             ("conv2a+conv3b plain", dict(plain=["conv2a", "conv3b"])),
             ("conv2a..conv3b plain", dict(plain=["conv2a", "conv2b", "conv3a", "conv3b"]))]
    print(f"# {H}x{W}; descriptor error against the fp32 twin: dense max | sampled max | dense rms   (worst over seeds 0..2; per seed in brackets: sampled)")
    for fam in fams:
        rows = {k: [] for k, _ in cands}
        for seed in range(3):
            sd = synth.make_state_dict(seed, None if fam == "default" else fam)
            img = synth.make_image(H, W, seed)
            x = tt.norm_rgb(torch.from_numpy(img)[None])
            with torch.no_grad():
                ref_tw = tt.Twin(sd)
                _, draw, _ = ref_tw.det_raw(x)
                ref = F.normalize(draw, dim=1)
                kp = tt.extract(ref_tw, img, topK=1024)["keypoints"]
                ref_s = eb.sample(ref, kp, H, W)
                for label, kw in cands:
                    _, d, _ = RelaxTwin(sd, **kw).det_raw(x)
                    d = F.normalize(d, dim=1)
                    rows[label].append((float((d - ref).abs().max()), float((eb.sample(d, kp, H, W) - ref_s).abs().max()),
                                        float((d - ref).pow(2).mean().sqrt())))
        print(f"## family {fam}")
        for label, _ in cands:
            r = np.array(rows[label])
            print(f"{label:26s} {r[:, 0].max():9.2e} {r[:, 1].max():9.2e} {r[:, 2].max():9.2e}   [{' '.join('%.2e' % v for v in r[:, 1])}]", flush=True)


if __name__ == "__main__":
    main()
