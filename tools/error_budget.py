#!/usr/bin/env python3
"""Per-layer error budget of the fp16 conv stack, on the CPU (oracle/torch_twin.py emulates the HIP path's rounding:
fp16 filters, fp16 activation storage, fp32 accumulate; head outputs fp32).

For each policy: max-abs error of the dense unit-norm descriptor map, of the descriptors sampled at the fp32 run's
key points, of the convPb logits, plus key-point set IoU -- all against the all-fp32 twin (which equals the C oracle
to 2e-6).  Usage:  python tools/error_budget.py [HxW ...]      (default 480x640)
Output is committed under profiles/ (r02_error_budget.txt).
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import torch_twin as tt   # noqa: E402
from sfd2_amd import synth            # noqa: E402

LAYERS = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b"] + \
         [f"conv4.{b}.conv{i}" for b in range(3) for i in (1, 2, 3)] + \
         ["convPa.0", "convPa.3", "convPb", "convDa.0", "convDa.3", "convDb", "ConvSta"]
HEADS = ("convPb", "convDb", "ConvSta")


def base_policy(act="f16", wt="f16"):
    pol = {}
    for n in LAYERS:   # x rounding only where the operand is not a stored activation (conv1a reads the fp32 image)
        pol[n] = tt.Policy(w=wt, x=act if n == "conv1a" else "f32", out="f32" if n in HEADS else act)
    pol["ConvSta"] = tt.Policy(w="f32", x="f32", out="f32")   # convsta_kernel keeps its 3x256 filters in fp32
    return pol


def with_(pol, **over):
    """over: layer -> (w, x, out) with None = keep."""
    q = {k: tt.Policy(v.w, v.x, v.out) for k, v in pol.items()}
    for name, (w, x, out) in over.items():
        name = name.replace("_", ".")
        p = q[name]
        q[name] = tt.Policy(w or p.w, x or p.x, out or p.out)
    return q


def run(sd, img, pol, ref=None, topK=1024):
    tw = tt.Twin(sd, pol)
    x = tt.norm_rgb(torch.from_numpy(img)[None])
    with torch.no_grad():
        logits, draw, sta = tw.det_raw(x)
        desc = F.normalize(draw, dim=1)
    ex = tt.extract(tw, img, topK=topK)
    out = {"logits": logits, "desc": desc, "draw": draw, "ex": ex}
    if ref is None:
        return out, None
    kp_ref = ref["ex"]["keypoints"]
    H, W = img.shape[1:]
    with torch.no_grad():   # this policy's descriptors sampled at the REFERENCE key points
        gx = torch.from_numpy(kp_ref[:, 0]).float() / (W / 2.0) - 1.0
        gy = torch.from_numpy(kp_ref[:, 1]).float() / (H / 2.0) - 1.0
        grid = torch.stack([gx, gy], 1).view(1, 1, -1, 2)
        d = F.grid_sample(desc, grid, mode="bilinear", align_corners=False)[0, :, 0].t()
        d = (d / d.norm(dim=1, keepdim=True)).double().numpy()
    a = {tuple(p) for p in ex["keypoints"].astype(int)}
    b = {tuple(p) for p in kp_ref.astype(int)}
    m = {"desc_dense": float((desc - ref["desc"]).abs().max()),
         "desc_sampled": float(np.abs(d - ref["ex"]["descriptors"]).max()),
         "logits": float((logits - ref["logits"]).abs().max()),
         "draw_rel": float((draw - ref["draw"]).abs().max() / ref["draw"].abs().max()),
         "iou": len(a & b) / max(1, len(a | b))}
    return out, m


def main():
    sizes = [tuple(int(v) for v in s.lower().split("x")) for s in sys.argv[1:]] or [(480, 640)]
    sd = synth.make_state_dict(0)
    X = "f16x2"
    for H, W in sizes:
        img = synth.make_image(H, W, 5)
        big = H * W > 1_000_000
        t0 = time.time()
        ref, _ = run(sd, img, None, topK=4096 if big else 1024)
        print(f"# {H}x{W}: fp32 twin {time.time() - t0:.1f}s; errors vs the fp32 twin; top-K {len(ref['ex']['keypoints'])}")
        print(f"{'policy':62s} {'desc_dense':>10s} {'desc_samp':>10s} {'desc_rms':>9s} {'logits':>9s} {'kp IoU':>7s}")
        base = base_policy()
        BB = LAYERS[:15]

        def show(name, pol):
            out, m = run(sd, img, pol, ref, topK=len(ref["ex"]["keypoints"]))
            r = float(((out["desc"] - ref["desc"]) ** 2).mean().sqrt())
            print(f"{name:62s} {m['desc_dense']:10.2e} {m['desc_sampled']:10.2e} {r:9.2e} {m['logits']:9.2e} {m['iou']:7.3f}", flush=True)

        print("## whole-network variants")
        show("f16 everywhere (throughput mode 'f16')", base)
        show("bf16 everywhere", base_policy("bf16", "bf16"))
        show("f16 activations, f32 filters", base_policy("f16", "f32"))
        show("f32 activations, f16 filters", base_policy("f32", "f16"))
        show("f16x2 (hi+lo fp16) filters and activations everywhere", base_policy(X, X))
        heads16 = dict(convPa_0=("f16", "f32", "f16"), convPa_3=("f16", "f32", "f16"), convPb=("f16", "f32", None),
                       convDa_0=("f16", "f32", "f16"), convDa_3=("f16", "f32", "f16"), convDb=("f16", "f32", None))
        pol = with_(base_policy("f32", "f32"), **heads16)
        pol["conv4.2.conv3"] = tt.Policy("f32", "f32", "f16")   # the heads read an fp16 copy of the backbone output
        show("backbone f32, both head branches f16", pol)
        show("descriptor branch f32 (convDa.0, convDa.3, convDb), rest f16",
             with_(base, convDa_0=("f32", None, "f32"), convDa_3=("f32", None, "f32"), convDb=("f32", None, None)))
        show("convDb hi+lo operands only", with_(base, convDa_3=(None, None, X), convDb=(X, None, None)))
        print("## first n backbone layers exact (filters f32, stored output f32), the rest f16")
        for n in ([1, 2, 3, 4, 5, 6, 9, 12, 15] if not big else [1, 2, 4, 6, 15]):
            over = {l.replace(".", "_"): ("f32", "f32", "f32") for l in BB[:n]}
            show(f"  n = {n:2d}  (through {BB[n - 1]})", with_(base, **over))
        print("## last n backbone layers exact, the rest f16")
        for n in ([3, 6, 9, 11, 13, 14] if not big else [9, 13]):
            over = {l.replace(".", "_"): ("f32", "f32", "f32") for l in BB[15 - n:]}
            show(f"  n = {n:2d}  (from {BB[15 - n]})", with_(base, **over))
        print("## hi+lo (f16x2) ladder on the early layers: w = filters, x = conv1a's image operand, out = stored activation")
        ladder = [
            ("conv1a x,w", dict(conv1a=(X, X, None))),
            ("conv1a x,w,out", dict(conv1a=(X, X, X))),
            ("  + conv1b w", dict(conv1a=(X, X, X), conv1b=(X, None, None))),
            ("  + conv1b out", dict(conv1a=(X, X, X), conv1b=(X, None, X))),
            ("  + conv2a w", dict(conv1a=(X, X, X), conv1b=(X, None, X), conv2a=(X, None, None))),
            ("  + conv2a out", dict(conv1a=(X, X, X), conv1b=(X, None, X), conv2a=(X, None, X))),
            ("  + conv2b w", dict(conv1a=(X, X, X), conv1b=(X, None, X), conv2a=(X, None, X), conv2b=(X, None, None))),
            ("  + conv2b out   [3 MFMA passes on stem, conv2a, conv2b; 2 on conv3a]",
             dict(conv1a=(X, X, X), conv1b=(X, None, X), conv2a=(X, None, X), conv2b=(X, None, X))),
            ("  + conv3a w,out [.. + 3 passes on conv3a, 2 on conv3b]",
             dict(conv1a=(X, X, X), conv1b=(X, None, X), conv2a=(X, None, X), conv2b=(X, None, X), conv3a=(X, None, X))),
            ("filters only hi+lo: conv1a .. conv2b (2 passes)", dict(conv1a=(X, X, None), conv1b=(X, None, None), conv2a=(X, None, None),
                                                                   conv2b=(X, None, None))),
            ("filters only hi+lo: conv1a .. conv3b (2 passes)", dict(conv1a=(X, X, None), conv1b=(X, None, None), conv2a=(X, None, None),
                                                                   conv2b=(X, None, None), conv3a=(X, None, None), conv3b=(X, None, None))),
            ("activations only hi+lo: conv1a .. conv2b (2 passes)", dict(conv1a=(None, X, X), conv1b=(None, None, X), conv2a=(None, None, X),
                                                                       conv2b=(None, None, X))),
        ]
        for name, over in ladder:
            show(name, with_(base, **over))


if __name__ == "__main__":
    main()
