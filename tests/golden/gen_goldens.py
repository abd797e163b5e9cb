#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE implementation (feixue94/sfd2,
mounted read-only at /root/reference) on seeded synthetic inputs and writes the
inputs' seeds + the reference's outputs as small .npz fixtures next to this file.

Run only in the authoring container (the reference does not exist on the GPU box):

    python tests/golden/gen_goldens.py

Nothing from the reference is copied: the fixtures hold numbers only. The three
shims below exist because the authoring image lacks torchvision / h5py / CUDA:
  * torchvision.transforms.{Compose,Normalize}  (nets/extractor.py:12,17)
  * h5py (imported at it_loc/matcher.py:12, used only in its main())
  * torch.Tensor.cuda -> identity (nets/extractor.py:106,205; it_loc/matcher.py:93-94)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SFD2_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

# ---------------------------------------------------------------- shims
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")


class _Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        # torchvision.transforms.functional.normalize: tensor.sub_(mean).div_(std)
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.clone().sub_(mean).div_(std)


class _Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


tvt.Normalize, tvt.Compose = _Normalize, _Compose
tv.transforms = tvt
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tvt
sys.modules["h5py"] = types.ModuleType("h5py")
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

from nets.sfd2 import ResSegNetV2  # noqa: E402
from nets import extractor as ref_ext  # noqa: E402
from hloc.matchers.nearest_neighbor import NearestNeighbor  # noqa: E402
from hloc.utils.parsers import names_to_pair  # noqa: E402
from it_loc.matcher import Matcher, confs as itloc_confs  # noqa: E402

from sfd2_amd import synth  # noqa: E402

# extract.py imports tools.dataloader (-> datasets, torchvision, ...) only for norm_RGB
_tools = types.ModuleType("tools")
_tdl = types.ModuleType("tools.dataloader")
_tdl.norm_RGB = ref_ext.norm_RGB
_tools.dataloader = _tdl
sys.modules["tools"] = _tools
sys.modules["tools.dataloader"] = _tdl
import extract as ref_extract  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def ref_model(seed=0):
    m = ResSegNetV2(outdim=128, require_stability=True).eval()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in synth.make_state_dict(seed).items()}
    m.load_state_dict(sd, strict=True)
    return m


def norm_rgb(img):
    return ref_ext.norm_RGB(torch.from_numpy(img))[None]


def subsample(t, n=2048):
    flat = t.detach().contiguous().view(-1)
    stride = max(1, flat.numel() // n)
    return flat[::stride][:n].numpy().copy(), stride


def gen_det(model, h, w, seed, tag):
    """G1/G2: det() outputs + strided samples of every intermediate activation."""
    img = synth.make_image(h, w, seed)
    x = norm_rgb(img)
    acts = {}
    hooks = []

    def add(name):
        def fn(mod, inp, out):
            acts[name] = out.detach().clone()
        return fn

    names = ["conv1a", "bn1b", "conv2a", "bn2b", "conv3a", "bn3b", "conv4.0", "conv4.1", "conv4.2",
             "conv4.0.bn1", "conv4.0.bn2", "convPa", "convDa", "convPb", "convDb", "ConvSta"]
    mods = dict(model.named_modules())
    for n in names:
        hooks.append(mods[n].register_forward_hook(add(n)))
    score, stab, desc = model.det(x)
    for hk in hooks:
        hk.remove()
    out = {"h": h, "w": w, "seed": seed,
           "score": score[0, 0].numpy(), "stability": stab[0, 0].numpy(), "desc": desc[0].numpy()}
    # heat map as used by the extractor (nets/extractor.py:137-141)
    heat = score
    if heat.shape[2] != h or heat.shape[3] != w:
        heat = torch.nn.functional.interpolate(heat, size=[h, w], mode="bilinear", align_corners=False)
    out["heat"] = (heat * stab)[0, 0].numpy()
    for n, a in acts.items():
        s, stride = subsample(a)
        out["act/" + n] = s
        out["act_shape/" + n] = np.array(a.shape)
        out["act_stride/" + n] = stride
    np.savez_compressed(os.path.join(HERE, f"det_{tag}.npz"), **out)
    print(f"det_{tag}: score {tuple(score.shape)} desc {tuple(desc.shape)}")


def gen_nms():
    """G3: simple_nms (nets/extractor.py:20-35) on random maps, a plateau map and
    a dense-peaks map; outputs stored sparsely (row-major nonzero index + value)."""
    out = {}
    rs = np.random.RandomState(7)
    cases = {
        "rand_61x83": rs.random_sample((61, 83)).astype(np.float32),
        "rand_128x160": rs.random_sample((128, 160)).astype(np.float32),
        # quantised -> many exact ties / plateaus
        "plateau_64x64": (np.floor(rs.random_sample((64, 64)) * 4) / 4).astype(np.float32),
        # mostly zero with sparse peaks, incl. peaks on the border
        "sparse_70x90": np.where(rs.random_sample((70, 90)) > 0.97, rs.random_sample((70, 90)), 0).astype(np.float32),
        "tiny_5x7": rs.random_sample((5, 7)).astype(np.float32),
    }
    for name, m in cases.items():
        r = ref_ext.simple_nms(torch.from_numpy(m)[None, None], 4)[0, 0].numpy()
        idx = np.flatnonzero(r)
        out[name + "/in"] = m
        out[name + "/idx"] = idx.astype(np.int64)
        out[name + "/val"] = r.reshape(-1)[idx]
    np.savez_compressed(os.path.join(HERE, "nms.npz"), **out)
    print("nms:", {k: len(v) for k, v in out.items() if k.endswith("/idx")})


def gen_extract(model, h, w, seed, topk, tag, keep_desc_all=False):
    """G4: extract_resnet_return (nets/extractor.py:97-338, no-mask branch)."""
    img = synth.make_image(h, w, seed)
    pred = ref_ext.extract_resnet_return(model, img=torch.from_numpy(img)[None], topK=topk,
                                         mask=None, conf_th=0.001, scales=[1.0])
    kp, sc, de = pred["keypoints"], pred["scores"], pred["descriptors"]
    assert kp.dtype == np.float64 and de.dtype == np.float64 and sc.dtype == np.float64
    # tie audit: the reference's own order among equal scores is implementation-defined
    n_ties = int((np.diff(sc) == 0).sum())
    # also record the candidate list after NMS+threshold (before border / top-K)
    x = norm_rgb(img)
    score, stab, desc = model.det(x)
    heat = score
    if heat.shape[2] != h or heat.shape[3] != w:
        heat = torch.nn.functional.interpolate(heat, size=[h, w], mode="bilinear", align_corners=False)
    heat = heat * stab
    nms = ref_ext.simple_nms(heat, 4)[0, 0].numpy()
    cand = np.flatnonzero(nms > 0.001)
    out = {"h": h, "w": w, "seed": seed, "topk": topk,
           "keypoints": kp.astype(np.float32), "scores": sc.astype(np.float32),
           "descriptors": de.astype(np.float16), "n_ties_in_selected": n_ties,
           "cand_idx": cand.astype(np.int64), "cand_val": nms.reshape(-1)[cand],
           "desc_norm_err": float(np.abs(np.linalg.norm(de, axis=1) - 1).max())}
    np.savez_compressed(os.path.join(HERE, f"extract_{tag}.npz"), **out)
    print(f"extract_{tag}: N={len(sc)} N0={len(cand)} ties={n_ties} min score {sc.min():.5f}")


def gen_extract_ms(model, h, w, seed, topk, scales, tag):
    """G4b: extract_resnet_return with a scale pyramid (nets/extractor.py:113-124,211-236,322-330)."""
    img = synth.make_image(h, w, seed)
    pred = ref_ext.extract_resnet_return(model, img=torch.from_numpy(img)[None], topK=topk,
                                         mask=None, conf_th=0.001, scales=list(scales))
    kp, sc, de = pred["keypoints"], pred["scores"], pred["descriptors"]
    out = {"h": h, "w": w, "seed": seed, "topk": topk, "scales": np.asarray(scales, dtype=np.float64),
           "keypoints": kp.astype(np.float32), "scores": sc.astype(np.float32),
           "descriptors": de.astype(np.float16)}
    np.savez_compressed(os.path.join(HERE, f"extract_ms_{tag}.npz"), **out)
    print(f"extract_ms_{tag}: N={len(sc)} scales={list(scales)}")


def make_mask(h, w, seed):
    """Synthetic instance mask (cv2 BGR uint8): 16-px blocks, ~40 % unlabelled (black), labels spread over the three bytes."""
    rs = np.random.RandomState(seed)
    gh, gw = (h + 15) // 16, (w + 15) // 16
    ids = rs.randint(1, 9, (gh, gw)) * (rs.random_sample((gh, gw)) > 0.4)
    pal = np.array([[0, 0, 0]] + [[(37 * i) % 256, (91 * i) % 7, i % 3] for i in range(1, 9)], dtype=np.uint8)
    return np.repeat(np.repeat(pal[ids], 16, axis=0), 16, axis=1)[:h, :w].copy()


def gen_extract_mask(model, h, w, seed, topk, tag):
    """G4c: the semantic-mask selection branch of extract_resnet_return (nets/extractor.py:240-319)."""
    for name in ("float", "int"):
        if not hasattr(np, name):           # the reference still spells np.float / np.int there
            setattr(np, name, {"float": float, "int": int}[name])
    img = synth.make_image(h, w, seed)
    mask = make_mask(h, w, seed + 1000)
    pred = ref_ext.extract_resnet_return(model, img=torch.from_numpy(img)[None], topK=topk, mask=mask, conf_th=0.001, scales=[1.0])
    out = {"h": h, "w": w, "seed": seed, "topk": topk, "mask": mask,
           "keypoints": pred["keypoints"].astype(np.float32), "scores": pred["scores"].astype(np.float32),
           "descriptors": pred["descriptors"].astype(np.float16), "labels": pred["labels"].astype(np.int32)}
    np.savez_compressed(os.path.join(HERE, f"extract_mask_{tag}.npz"), **out)
    print(f"extract_mask_{tag}: N={len(pred['scores'])} labelled={int((pred['labels'] != 0).sum())}")


def gen_extract_spp(model, h, w, seed, conf_th, tag):
    """G5: extract.py nms_fast (:17-84) and extract_spp_feats_singlescale (:205-277)."""
    img = synth.make_image(h, w, seed)
    x = norm_rgb(img)
    pts, desc, scores, desc_full, heat = ref_extract.extract_spp_feats_singlescale(model, x, conf_th=conf_th)
    out = {"h": h, "w": w, "seed": seed, "conf_th": conf_th, "pts": pts, "desc": desc.astype(np.float16),
           "n_ties": int((np.diff(pts[:, 2]) == 0).sum()), "heat": heat.astype(np.float32)}
    # nms_fast alone on random integer corners
    rs = np.random.RandomState(17)
    hh, ww, n = 60, 80, 900
    lin = rs.permutation(hh * ww)[:n]
    corners = np.stack([lin % ww, lin // ww, rs.random_sample(n) + 0.01]).astype(np.float64)
    o, inds = ref_extract.nms_fast(corners, hh, ww, 4)
    out.update({"nf/corners": corners, "nf/h": hh, "nf/w": ww, "nf/out": o, "nf/inds": inds.astype(np.int64)})
    np.savez_compressed(os.path.join(HERE, f"extract_spp_{tag}.npz"), **out)
    print(f"extract_spp_{tag}: N={len(pts)} ties={out['n_ties']} nms_fast kept {o.shape[1]} of {n}")


def gen_extract_spp_ms(model, h, w, seed, conf_th, min_size, tag):
    """extract.py extrat_spp_feats_multiscale (:87-201), scale_f = 1.2 as extract_spp_return calls it (:295-297)."""
    img = synth.make_image(h, w, seed)
    x = norm_rgb(img)
    pts, desc, scores = ref_extract.extrat_spp_feats_multiscale(model, x, conf_th=conf_th, scale_f=1.2, min_size=min_size, max_size=9999)
    out = {"h": h, "w": w, "seed": seed, "conf_th": conf_th, "min_size": min_size, "pts": pts, "desc": desc.astype(np.float32)}
    np.savez_compressed(os.path.join(HERE, f"extract_spp_ms_{tag}.npz"), **out)
    print(f"extract_spp_ms_{tag}: N={len(pts)}")


def gen_matchers():
    """G6: hloc NearestNeighbor (hloc/matchers/nearest_neighbor.py:27-57) with the
    NNM / ONN / NNR confs (hloc/match_features.py:20-45) + a ratio-test conf, and
    it_loc Matcher nnm / nnr (it_loc/matcher.py:85-194)."""
    out = {}
    for tag, n0, n1, s0, s1 in [("a", 1024, 777, 1, 2), ("b", 300, 512, 3, 4)]:
        d0 = synth.make_descriptors(n0, seed=s0)
        d1 = synth.make_descriptors(n1, seed=s1)
        # make ~half of d1 noisy copies of d0 rows so that real matches exist
        rs = np.random.RandomState(50 + s0)
        k = min(n0, n1) // 2
        src = rs.permutation(n0)[:k]
        dst = rs.permutation(n1)[:k]
        sigma = (0.02 + 0.10 * rs.random_sample((k, 1))).astype(np.float32)
        noisy = d0[src] + sigma * rs.standard_normal((k, 128)).astype(np.float32)
        noisy /= np.linalg.norm(noisy, axis=1, keepdims=True)
        d1[dst] = noisy.astype(np.float32)
        out[f"{tag}/d0"], out[f"{tag}/d1"] = d0, d1
        hl = {
            "NNM": {"do_mutual_check": True, "distance_threshold": None},
            "ONN": {"do_mutual_check": False, "distance_threshold": None},
            "NNR": {"do_mutual_check": True, "distance_threshold": 0.9},
            "RATIO": {"do_mutual_check": True, "ratio_threshold": 0.8, "distance_threshold": None},
            "RATIO_DIST": {"do_mutual_check": False, "ratio_threshold": 0.9, "distance_threshold": 0.7},
        }
        data = {"descriptors0": torch.from_numpy(d0.T.copy())[None], "descriptors1": torch.from_numpy(d1.T.copy())[None]}
        sim = (torch.from_numpy(d0).double() @ torch.from_numpy(d1).double().T)
        top2 = sim.topk(2, dim=1)[0]
        out[f"{tag}/row_gap_min"] = float((top2[:, 0] - top2[:, 1]).min())
        top2c = sim.T.topk(2, dim=1)[0]
        out[f"{tag}/col_gap_min"] = float((top2c[:, 0] - top2c[:, 1]).min())
        for name, conf in hl.items():
            pred = NearestNeighbor(conf).eval()(data)
            out[f"{tag}/hloc/{name}/matches0"] = pred["matches0"][0].numpy()
            out[f"{tag}/hloc/{name}/scores0"] = pred["matching_scores0"][0].numpy()
        for name in ("NNM", "NNR"):
            pred = Matcher(itloc_confs[name]).eval()({"descriptors0": d0.astype(np.float64),
                                                      "descriptors1": d1.astype(np.float64)})
            out[f"{tag}/itloc/{name}/matches0"] = np.asarray(pred["matches0"]).astype(np.int64)
            out[f"{tag}/itloc/{name}/scores0"] = np.asarray(pred["matching_scores0"]).astype(np.float64)
        # label-aware matcher (it_loc/matcher.py:239-297); the reference still spells np.int there
        if not hasattr(np, "int"):
            np.int = int
        rl = np.random.RandomState(90 + s0)
        l0 = rl.randint(0, 6, n0).astype(np.int32)
        l1 = rl.randint(0, 6, n1).astype(np.int32)
        l1[dst] = l0[src]                     # the planted pairs mostly share their label
        flip = rl.random_sample(k) < 0.2
        l1[dst[flip]] = rl.randint(0, 6, int(flip.sum()))
        pred = Matcher({"output": "NNML", "model": {"name": "nnml"}}).eval()(
            {"descriptors0": d0.astype(np.float64), "descriptors1": d1.astype(np.float64), "labels0": l0, "labels1": l1})
        out[f"{tag}/labels0"], out[f"{tag}/labels1"] = l0, l1
        out[f"{tag}/itloc/NNML/matches0"] = np.asarray(pred["matches0"]).astype(np.int64)
        out[f"{tag}/itloc/NNML/scores0"] = np.asarray(pred["matching_scores0"]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "matchers.npz"), **out)
    print("matchers:", {k: (int((v >= 0).sum()) if v.ndim else v) for k, v in out.items() if k.endswith("matches0")})


def gen_host():
    """G7: host-side formulas of the drivers (no device arithmetic)."""
    out = {}
    # extract_localization.py:260-263
    kp = np.array([[4.0, 4.0], [100.0, 37.0], [1595.0, 1058.0]])
    original_size = np.array([4032, 2680])
    size = np.array([1600, 1063])
    scales = (original_size / size).astype(np.float32)
    out["rescale/kp"], out["rescale/orig"], out["rescale/size"] = kp, original_size, size
    out["rescale/out"] = (kp + .5) * scales[None] - .5
    # hloc/match_features.py:114,118 casts
    m = np.array([-1, 0, 5, 4095, 40000], dtype=np.int64)
    out["cast/m_in"] = m
    out["cast/m_out"] = torch.from_numpy(m).short().numpy()
    s = np.array([0.0, 0.5, 0.99951172, 1.0, 0.123456789], dtype=np.float32)
    out["cast/s_in"] = s
    out["cast/s_out"] = torch.from_numpy(s).half().numpy()
    pairs = [("query/day/nexus4/IMG_1.jpg", "db/1045.jpg"), ("a.jpg", "b/c/d.png")]
    out["pairs/in"] = np.array(pairs)
    out["pairs/out"] = np.array([names_to_pair(a, b) for a, b in pairs])
    np.savez_compressed(os.path.join(HERE, "host.npz"), **out)
    print("host ok")


if __name__ == "__main__":
    torch.set_num_threads(8)
    model = ref_model(0)
    gen_extract_ms(model, 96, 128, 21, 150, [1.0, 0.5], "96x128_k150")
    gen_extract_ms(model, 100, 130, 22, -1, [1.2, 1.0, 0.6], "100x130_all")
    if len(sys.argv) > 1 and sys.argv[1] == "ms":
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mask":
        gen_extract_mask(model, 96, 128, 21, 120, "96x128_k120")      # topK <= labelled
        gen_extract_mask(model, 96, 128, 21, 180, "96x128_k180")      # labelled < topK < all
        gen_extract_mask(model, 100, 130, 22, 5000, "100x130_k5000")  # topK >= all
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sppms":
        gen_extract_spp_ms(model, 96, 128, 21, 0.02, 40, "96x128")
        gen_extract_spp_ms(model, 100, 130, 22, 0.05, 64, "100x130")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "matchers":
        gen_matchers()
        sys.exit(0)
    gen_det(model, 64, 96, 11, "64x96")
    gen_det(model, 100, 130, 12, "100x130")
    gen_nms()
    gen_extract(model, 96, 128, 21, 200, "96x128_k200")
    gen_extract(model, 100, 130, 22, -1, "100x130_all")
    gen_extract(model, 480, 640, 0, 1024, "480x640_k1024")
    gen_extract_spp(model, 96, 128, 21, 0.02, "96x128")
    gen_extract_spp_ms(model, 96, 128, 21, 0.02, 40, "96x128")
    gen_extract_spp_ms(model, 100, 130, 22, 0.05, 64, "100x130")
    gen_matchers()
    gen_host()
