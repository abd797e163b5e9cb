#!/usr/bin/env python3
"""What would the compensated mode's descriptors look like with fp6 correction operands?  CPU study on the torch twin
(oracle/torch_twin.py): the f16c arithmetic of DESIGN.md section 3 -- y = conv(hi_x, hi_w) + conv(q(lo_x), q(w)) + conv(q(x), q(lo_w)),
fp32 accumulate, stored activations = (hi, q(lo), q(x)) -- with the quantiser q of the correction factors as a parameter:
  e4m3        today's corr units (4 significant bits, own exponent per value)
  fp6w        filters' correction factors as e2m3 with ONE power-of-two scale per output channel, activations e4m3
              (v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 x fp6 operands: 32 ns instead of 37-41 per instruction)
  fp6         both sides e2m3; activations with a shared power-of-two scale per pixel and 32 channels (fp6 x fp6: 21.5 ns)
ResBlocks as option rb_inner = 2 (t1 / t2 plain fp16, filters as hi + fp16 residual, skip path compensated), heads plain fp16.
Errors against the all-fp32 twin: dense unit-norm descriptor map and descriptors sampled at the fp32 run's key points.
    python tools/error_budget_fp6.py [HxW ...]        (default 480x640)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import torch_twin as tt   # noqa: E402
from sfd2_amd import synth            # noqa: E402


def f16(t):
    return t.to(torch.float16).to(torch.float32)


def q_e4m3(v, pre):
    """e4m3 of v * pre (normals 2^-6 .. 448, subnormal step 2^-9, saturating), returned at v's scale."""
    a = (v * pre).abs().clamp(max=448.0)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=-6.0)
    step = torch.exp2(e - 3.0)
    q = torch.round(a / step) * step
    return torch.sign(v) * q.clamp(max=448.0) / pre


def q_e2m3(a_scaled):
    """e2m3 grid on |a| <= 7.5: normals 1 .. 7.5 (3 mantissa bits), subnormal step 0.125."""
    a = a_scaled.abs().clamp(max=7.5)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=0.0, max=2.0)
    step = torch.exp2(e - 3.0)
    return torch.sign(a_scaled) * (torch.round(a / step) * step).clamp(max=7.5)


def pow2_scale(amax):
    return torch.exp2(torch.ceil(torch.log2((amax / 7.5).clamp(min=1e-30))))


def q_fp6_act(v, ref):
    """v, ref [1,C,H,W]: e2m3 of v with a power-of-two scale per pixel and 32 channels taken from |ref| (the block's activations)."""
    n, c, h, w = v.shape
    g = ref.abs().view(n, c // 32, 32, h, w).amax(dim=2, keepdim=True)
    s = pow2_scale(g).expand(n, c // 32, 32, h, w).reshape(n, c, h, w)
    return q_e2m3(v / s) * s


def q_fp6_w(v, ref):
    """v, ref [Cout,Cin,k,k]: e2m3 with one power-of-two scale per output channel taken from |ref|."""
    s = pow2_scale(ref.abs().flatten(1).amax(dim=1)).view(-1, 1, 1, 1)
    return q_e2m3(v / s) * s


class CompTwin(tt.Twin):
    def __init__(self, sd, mode):
        super().__init__(sd)
        self.mode = mode

    def layer(self, name, x, conv, bn, stride=1, relu=True, groups=1, residual=None):
        w = self.sd[conv + ".weight"]
        k = w.shape[-1]
        cv = lambda a, b: F.conv2d(a, b, None, stride=stride, padding=k // 2, groups=groups)   # noqa: E731
        wh = f16(w)
        wl = w - wh
        inner = name.startswith("conv4.") and (name.endswith("conv2") or name.endswith("conv3"))
        head = name.startswith(("convP", "convD", "ConvSta"))
        if name == "ConvSta":
            y = cv(f16(x), w)
        elif head:
            y = cv(f16(x), wh)
        elif name == "conv1a":                   # image and filters as hi + lo fp16, three passes
            xh = f16(x)
            y = cv(xh, wh) + cv(xh, f16(wl)) + cv(f16(x - xh), wh)
        elif inner:                              # plain fp16 input, filters hi + fp16 residual
            y = cv(f16(x), wh + f16(wl * 2048.0) / 2048.0)
        else:
            xh = f16(x)
            xl = x - xh
            if self.mode == "fp6" and x.shape[1] % 32 == 0:
                qxl, qx = q_fp6_act(xl * 2048.0, x) / 2048.0, q_fp6_act(x, x)
            else:
                qxl, qx = q_e4m3(xl, 512.0), q_e4m3(x, 0.25)
            if self.mode in ("fp6", "fp6w"):
                qw, qwl = q_fp6_w(w, w), q_fp6_w(wl * 2048.0, w) / 2048.0
            else:
                amax = w.abs().max()
                pre = 224.0 / amax               # (the layer's scale byte: the largest filter sits in the top binade)
                qw, qwl = q_e4m3(w, pre), q_e4m3(wl, pre * 2048.0)
            y = cv(xh, wh) + cv(qxl, qw) + cv(qx, qwl)
        a, s = self._fold(conv, bn)
        y = y * a.view(1, -1, 1, 1) + s.view(1, -1, 1, 1)
        if residual is not None:                 # skip path: hi + the corr unit's residual
            rh = f16(residual)
            rl = residual - rh
            y = y + rh + (q_fp6_act(rl * 2048.0, residual) / 2048.0 if self.mode == "fp6" else q_e4m3(rl, 512.0))
        if relu:
            y = F.relu(y)
        if name.startswith("conv4.") and (name.endswith("conv1") or name.endswith("conv2")):
            y = f16(y)                           # t1 / t2 are plain fp16 (rb_inner = 2)
        return y


def sample(desc, kp, H, W):
    gx = torch.from_numpy(kp[:, 0]).float() / (W / 2.0) - 1.0
    gy = torch.from_numpy(kp[:, 1]).float() / (H / 2.0) - 1.0
    d = F.grid_sample(desc, torch.stack([gx, gy], 1).view(1, 1, -1, 2), mode="bilinear", align_corners=False)[0, :, 0].t()
    return d / d.norm(dim=1, keepdim=True)


def main():
    sizes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(480, 640)]
    sd = synth.make_state_dict(0)
    for H, W in sizes:
        img = synth.make_image(H, W, 0)
        x = tt.norm_rgb(torch.from_numpy(img)[None])
        with torch.no_grad():
            ref_tw = tt.Twin(sd)
            _, draw, _ = ref_tw.det_raw(x)
            ref = F.normalize(draw, dim=1)
            kp = tt.extract(ref_tw, img, topK=1024 if H * W < 1000000 else 4096)["keypoints"]
            ref_s = sample(ref, kp, H, W)
            print(f"# {H}x{W}: errors against the fp32 twin, {len(kp)} key points")
            print(f"{'correction operands':44s} desc_dense  desc_sampled  desc_rms")
            for mode, label in (("e4m3", "e4m3 both sides (today)"), ("fp6w", "filters e2m3 per-channel scale, act e4m3"),
                                ("fp6", "both e2m3 (act: scale per pixel x 32 ch)")):
                _, d, _ = CompTwin(sd, mode).det_raw(x)
                d = F.normalize(d, dim=1)
                print(f"{label:44s} {float((d - ref).abs().max()):9.2e}  {float((sample(d, kp, H, W) - ref_s).abs().max()):11.2e}  "
                      f"{float((d - ref).pow(2).mean().sqrt()):9.2e}", flush=True)


if __name__ == "__main__":
    main()
