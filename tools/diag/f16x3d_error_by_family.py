"""Descriptor error of precision 'f16x3d' (strict backbone + detector, fp16 descriptor branch) against the library's own f32 by synthetic weight family, 480x640 top-1024."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sfd2_amd import synth
from sfd2_amd.extractor import extract_resnet_return
from sfd2_amd.model import ResSegNetV2
def model(sd, prec):
    m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval(); m.load_state_dict(sd); m.cuda(0); return m
for fam in (None, "student", "calibrated", "biased", "dead", "smallvar"):
    worst = 0.0
    for seed in range(3):
        sd = synth.make_state_dict(seed, family=fam)
        img = synth.make_image(480, 640, 40 + seed)
        a = extract_resnet_return(model(sd, "f32"), img[None], conf_th=0.001, topK=1024, scales=[1.0])
        b = extract_resnet_return(model(sd, "f16x3d"), img[None], conf_th=0.001, topK=1024, scales=[1.0])
        ka = {(float(x), float(y)): i for i, (x, y) in enumerate(a["keypoints"])}; kb = {(float(x), float(y)): i for i, (x, y) in enumerate(b["keypoints"])}
        c = sorted(set(ka) & set(kb))
        d = float(np.abs(a["descriptors"][[ka[k] for k in c]] - b["descriptors"][[kb[k] for k in c]]).max())
        worst = max(worst, d)
        print(f"{fam or 'default'} seed {seed}: {len(c)}/{len(ka)} common key points, descriptors {d:.2e}", flush=True)
    print(f"{fam or 'default'}: worst {worst:.2e}")
