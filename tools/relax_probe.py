#!/usr/bin/env python3
"""What the load-time self-check of SFD2_PREC_F16C finds with option "c3b_plain" (round 6): per weight family and seed the probe errors
(options as set, with conv3b plain), what the context ends up running, and the descriptor error of a 480x640 top-1024 extraction against
the oracle on the chosen options.   python tools/relax_probe.py [seeds]   -> one line per (family, seed)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sfd2_amd import synth                      # noqa: E402
from sfd2_amd.model import ResSegNetV2          # noqa: E402
from sfd2_amd.extractor import extract_resnet_return   # noqa: E402
import oracle.oracle as orc                     # noqa: E402  (checker only)


def desc_err(m, sd, h=480, w=640, k=1024, seed=41):
    img = synth.make_image(h, w, seed)
    got = extract_resnet_return(m, img[None], conf_th=0.001, topK=k, scales=[1.0])
    want = orc.extract_resnet_return(sd, img, conf_th=0.001, topK=k)
    a = {(float(x), float(y)): i for i, (x, y) in enumerate(got["keypoints"])}
    b = {(float(x), float(y)): i for i, (x, y) in enumerate(want["keypoints"])}
    common = sorted(set(a) & set(b))
    ia = np.array([a[c] for c in common]); ib = np.array([b[c] for c in common])
    d = got["descriptors"][ia] - np.asarray(want["descriptors"])[ib]
    return float(np.abs(d).max()), float(np.sqrt((d * d).mean())), len(common) / max(1, len(set(a) | set(b)))


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for fam in (None, "student", "calibrated", "biased", "dead", "smallvar"):
        for seed in range(seeds):
            sd = synth.make_state_dict(seed, family=fam)
            res = []
            for plain in (-1, 0):
                m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
                m.cuda(0)
                m.context.set_option("c3b_plain", plain)
                m.load_state_dict(sd)
                st = m.context.margin_status()
                e, rms, iou = desc_err(m, sd)
                res.append((st, e, rms, iou))
            st = res[0][0]
            print(f"{fam or 'default':10s} seed {seed}: probe as set {st['errors']['as set']:.2e}, with c3b_plain {st['error_with_c3b_plain']:.2e} -> running '{st['running']}', "
                  f"c3b_plain {int(st['c3b_plain'])};  extraction vs oracle: auto {res[0][1]:.2e} (rms {res[0][2]:.2e}, IoU {res[0][3]:.4f}) | "
                  f"c3b_plain=0 '{res[1][0]['running']}' {res[1][1]:.2e} (rms {res[1][2]:.2e}, IoU {res[1][3]:.4f})", flush=True)


if __name__ == "__main__":
    main()
