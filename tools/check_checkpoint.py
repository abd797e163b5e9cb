#!/usr/bin/env python3
"""First contact with a checkpoint nobody has run through this library yet (the reference's 20220810_ressegnetv2_wapv2_ce_sd2mfsf_uspg.pth is not
shipped with it: /root/reference/.MISSING_LARGE_BLOBS).  Product only -- the reference for every number is the library's own SFD2_PREC_F32 mode, whose
arithmetic is an fp32 FMA chain:

    python tools/check_checkpoint.py weights.pth [image.jpg ...] [--size 1600x1200] [--topk 4096]

Prints: what the state_dict holds against what the loader takes (missing / unexpected names), the load-time self-check of f16c (probe error against f32,
the accuracy options it turned on), the activation exponents the range calibration chose, and per precision mode -- on the given images, else on
synthetic ones -- the descriptor error and key-point agreement against f32, the range status (saturated / low tensors, fallbacks) and ms per extract."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def load_sd(path):
    import torch
    sd = torch.load(path, map_location="cpu")
    sd = sd["model"] if isinstance(sd, dict) and isinstance(sd.get("model"), dict) else sd
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}


def images_of(args):
    from sfd2_amd import synth
    W, H = (int(v) for v in args.size.split("x"))
    out = []
    for p in args.images:
        from sfd2_amd.extract_localization import _read_rgb_u8, resized_shape
        u8 = _read_rgb_u8(p)
        h, w = u8.shape[:2]
        out.append((os.path.basename(p), u8, resized_shape(w, h, max(W, H), False)))
    if not out:
        for s in range(args.n_synth):
            out.append((f"synthetic seed {s}", (synth.make_image(H, W, 300 + s).transpose(1, 2, 0) * 255).astype(np.uint8).copy(), (W, H)))
    return out


def feed(model, u8, resize):
    from sfd2_amd.extract_localization import preprocess
    h, w = u8.shape[:2]
    return preprocess(model, u8, resize if tuple(resize) != (w, h) else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("weights")
    ap.add_argument("images", nargs="*")
    ap.add_argument("--size", default="1600x1200", help="synthetic image size / resize_max of real images (the longer side)")
    ap.add_argument("--topk", type=int, default=4096)
    ap.add_argument("--n-synth", type=int, default=2)
    ap.add_argument("--modes", default="f16x3,f16x3d,f16c,f16")
    args = ap.parse_args()

    from sfd2_amd import synth
    from sfd2_amd.extractor import extract_resnet_return
    from sfd2_amd.model import ResSegNetV2
    sd = load_sd(args.weights)
    expect = set(synth.make_state_dict(0).keys())
    have = set(sd.keys())
    print(f"state_dict: {len(have)} tensors; missing {sorted(expect - have)[:8]}{' ...' if len(expect - have) > 8 else ''}; "
          f"not used by the inference path: {len(have - expect)} (e.g. {sorted(have - expect)[:4]})")
    stab = "ConvSta.weight" in sd
    imgs = images_of(args)

    def model_of(prec):
        m = ResSegNetV2(outdim=128, require_stability=stab, precision=prec).eval()
        m.load_state_dict(sd, strict=False)
        m.cuda(0)
        return m

    ref_m = model_of("f32")
    refs = [extract_resnet_return(ref_m, feed(ref_m, u8, rs), conf_th=0.001, topK=args.topk, scales=[1.0]) for _, u8, rs in imgs]
    exps, maxima = ref_m.context.act_exponents()
    print("activation exponents (stored tensor = value x 2^e), calibration maxima:", " ".join(f"{int(e):+d}/{float(v):.3g}" for e, v in zip(exps, maxima)))
    for prec in args.modes.split(","):
        m = model_of(prec)
        if prec == "f16c":
            st = m.context.margin_status()
            print(f"f16c self-check: probe error {st['errors']}, running '{st['running']}' (target {st['target']:.1e}); conv3b without correction chunks "
                  f"(c3b_plain): probe with it {st['error_with_c3b_plain']:.2e} -> {'on' if st['c3b_plain'] else 'off'}")
        worst_d, worst_iou, same_rank = 0.0, 1.0, []
        for (name, u8, rs), ref in zip(imgs, refs):
            got = extract_resnet_return(m, feed(m, u8, rs), conf_th=0.001, topK=args.topk, scales=[1.0])
            a = {(float(x), float(y)): i for i, (x, y) in enumerate(got["keypoints"])}
            b = {(float(x), float(y)): i for i, (x, y) in enumerate(ref["keypoints"])}
            common = sorted(set(a) & set(b))
            iou = len(common) / max(1, len(set(a) | set(b)))
            ia = np.array([a[c] for c in common], dtype=np.int64); ib = np.array([b[c] for c in common], dtype=np.int64)
            d = float(np.abs(got["descriptors"][ia] - ref["descriptors"][ib]).max()) if common else float("nan")
            worst_d, worst_iou = max(worst_d, d), min(worst_iou, iou)
            same_rank.append(float((ia == ib).mean()) if common else 0.0)
        x = feed(m, imgs[0][1], imgs[0][2])
        extract_resnet_return(m, x, conf_th=0.001, topK=args.topk, scales=[1.0])
        t0 = time.perf_counter()
        for _ in range(10):
            extract_resnet_return(m, x, conf_th=0.001, topK=args.topk, scales=[1.0])
        ms = (time.perf_counter() - t0) / 10 * 1e3
        rs_ = m.range_status() if hasattr(m, "range_status") else {}
        print(f"{prec:7s}: descriptors vs f32 <= {worst_d:.2e}, key-point set IoU >= {worst_iou:.4f}, at the same rank {min(same_rank):.3f}; "
              f"saturated {rs_.get('saturated')}, low {rs_.get('low')}, fallbacks {rs_.get('fallbacks')}; {ms:.2f} ms per synchronous extract")
    print("tolerances the tests assert on the synthetic weights: f16x3 2e-5 (list equal up to near-ties), f16x3d 1e-3 with f16x3's key points, f16c 1e-3 / IoU 0.985, f16 3e-3 / 0.93")


if __name__ == "__main__":
    main()
