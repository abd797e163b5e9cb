#!/usr/bin/env python3
"""What the accuracy knobs of SFD2_PREC_F16C buy on the weight family with the thinnest margin (Student-t filters reach
9.3e-4 of the 1e-3 tolerance with the defaults).  For five seeds of synth.make_state_dict(family="student") at 480x640 / top-1024: descriptor
error against the fp32 oracle and ms per extract for the default options, `comp_heads` = 1, `rb_inner` = 0, and both.
    python tools/margin_knobs.py        (one table on stdout; committed under profiles/)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc          # noqa: E402  (tools/ are test infrastructure: the oracle is the checker here)
from sfd2_amd import synth                # noqa: E402
from sfd2_amd.extractor import extract_resnet_return   # noqa: E402
from sfd2_amd.model import ResSegNetV2    # noqa: E402

H, W, K = 480, 640, 1024
KNOBS = [("defaults", {}), ("comp_heads=1", {"comp_heads": 1}), ("rb_inner=0", {"rb_inner": 0}), ("comp_heads=1 rb_inner=0", {"comp_heads": 1, "rb_inner": 0})]


def errors(got, want):
    a = {(float(x), float(y)): i for i, (x, y) in enumerate(got["keypoints"])}
    b = {(float(x), float(y)): i for i, (x, y) in enumerate(want["keypoints"])}
    common = sorted(set(a) & set(b))
    ia = np.array([a[k] for k in common]); ib = np.array([b[k] for k in common])
    d = got["descriptors"][ia] - np.asarray(want["descriptors"], dtype=np.float64)[ib]
    return len(common) / max(1, len(set(a) | set(b))), float(np.abs(d).max()), float(np.sqrt((d ** 2).mean()))


rows = {k: [] for k, _ in KNOBS}
for seed in range(5):
    sd = synth.make_state_dict(seed, family="student")
    img = synth.make_image(H, W, 40 + seed)
    want = orc.extract_resnet_return(sd, img, conf_th=0.001, topK=K)
    for name, opts in KNOBS:
        m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval()
        m.cuda(0)
        for k, v in opts.items():
            m.context.set_option(k, v)
        m.load_state_dict(sd)
        got = extract_resnet_return(m, img[None], conf_th=0.001, topK=K)
        for _ in range(3):
            extract_resnet_return(m, img[None], conf_th=0.001, topK=K)
        t0 = time.perf_counter()
        for _ in range(10):
            extract_resnet_return(m, img[None], conf_th=0.001, topK=K)
        ms = (time.perf_counter() - t0) / 10 * 1e3
        iou, dmax, drms = errors(got, want)
        rows[name].append((dmax, drms, iou, ms))
print(f"f16c accuracy knobs on the Student-t weight family, {W}x{H} top-{K}, five seeds (descriptor error against the fp32 oracle; synchronous extract incl. PCIe)")
print(f"{'options':28s} {'max over seeds':>14s} {'per seed max':>52s} {'rms':>10s} {'IoU min':>8s} {'ms/extract':>10s}")
for name, _ in KNOBS:
    r = rows[name]
    print(f"{name:28s} {max(x[0] for x in r):14.2e} {' '.join(f'{x[0]:.2e}' for x in r):>52s} {np.mean([x[1] for x in r]):10.2e} {min(x[2] for x in r):8.3f} {np.mean([x[3] for x in r]):10.3f}")
