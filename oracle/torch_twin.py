"""TORCH-CPU TWIN (test infrastructure, NOT product code).

A second CPU restatement of ResSegNetV2.det / extract_resnet_return / the hloc NNM matcher, written here from
SURVEY.md section 8a over the synthetic state_dict with stock torch ops (F.conv2d / max_pool2d / grid_sample), i.e. the
same third-party arithmetic the reference itself runs on (SURVEY 8c last row).  Two uses:

  * bench.py's `cpu_baseline` leg: this is what "the reference's CPU path" costs on the GPU box's host cores
    (oneDNN convolutions, all physical cores) -- the C oracle next to it is a naive OpenMP loop nest.
  * tools/error_budget.py: every layer takes a `Policy` that rounds its filters / input activations the way the HIP
    fp16 path does (fp16 operands, fp32 accumulate, fp16 activation storage), so the per-layer error budget of a
    mixed-precision mode can be explored on the CPU before a kernel is written.

Reference lines followed: nets/sfd2.py:25-55 (ResBlock), :58-95 (conv/BN), :313-354 (det), nets/extractor.py:14-35,
:97-338, hloc/matchers/nearest_neighbor.py:6-57.  Only tests/, tools/error_budget.py, smoke() and bench.py import it.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
RGB_MEAN = (0.485, 0.456, 0.406)
RGB_STD = (0.229, 0.224, 0.225)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class Policy:
    """How one layer's operands are rounded.  'f32' = untouched, 'f16' / 'bf16' = round to that type,
    'f16x2' = hi + lo pair of fp16 (22 mantissa bits: what a split-operand MFMA sees)."""

    def __init__(self, w="f32", x="f32", out="f32"):
        self.w, self.x, self.out = w, x, out


def rnd(t, mode):
    if mode == "f32":
        return t
    if mode == "f16":
        return t.to(torch.float16).to(torch.float32)
    if mode == "bf16":
        return t.to(torch.bfloat16).to(torch.float32)
    if mode == "f16x2":
        hi = t.to(torch.float16).to(torch.float32)
        lo = (t - hi).to(torch.float16).to(torch.float32)
        return hi + lo
    raise ValueError(mode)


F32 = Policy()


class Twin:
    """det() over a state_dict of numpy arrays; policies: {layer name: Policy}, default for the rest."""

    def __init__(self, sd, policies=None, default=F32):
        self.sd = {k: _t(v) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
        self.pol = policies or {}
        self.default = default
        self.taps = None

    def _p(self, name):
        return self.pol.get(name, self.default)

    def _fold(self, conv, bn):
        """scale/shift of the folded conv bias + BatchNorm(eval) (nets/sfd2.py:58-65): applied in fp32 after the conv,
        never baked into the filters -- the HIP epilogue does the same."""
        sd = self.sd
        w = sd[conv + ".weight"]
        cout = w.shape[0]
        b = sd.get(conv + ".bias", torch.zeros(cout))
        if bn is None:
            return torch.ones(cout), b
        inv = 1.0 / torch.sqrt(sd[bn + ".running_var"] + BN_EPS)
        a = sd[bn + ".weight"] * inv if (bn + ".weight") in sd else inv
        beta = sd.get(bn + ".bias", torch.zeros(cout))
        return a, beta + (b - sd[bn + ".running_mean"]) * a

    def layer(self, name, x, conv, bn, stride=1, relu=True, groups=1, residual=None):
        p = self._p(name)
        w = rnd(self.sd[conv + ".weight"], p.w)
        xin = rnd(x, p.x)
        k = w.shape[-1]
        y = F.conv2d(xin, w, None, stride=stride, padding=k // 2, groups=groups)
        a, s = self._fold(conv, bn)
        y = y * a.view(1, -1, 1, 1) + s.view(1, -1, 1, 1)
        if residual is not None:
            y = y + rnd(residual, p.x)
        if relu:
            y = F.relu(y)
        y = rnd(y, p.out)
        if self.taps is not None:
            self.taps[name] = y
        return y

    def det_raw(self, x):
        """x [1,3,H,W] normalised.  Returns (convPb logits [1,65,H8,W8], convDb raw [1,128,H4,W4], ConvSta [1,3,H4,W4])."""
        L = self.layer
        o = L("conv1a", x, "conv1a.0", "conv1a.1")
        o = L("conv1b", o, "conv1b.0", "bn1b.0", stride=2)
        o = L("conv2a", o, "conv2a.0", "conv2a.1")
        o = L("conv2b", o, "conv2b.0", "bn2b.0", stride=2)
        o = L("conv3a", o, "conv3a.0", "conv3a.1")
        o = L("conv3b", o, "conv3b.0", "bn3b.0")
        for b in range(3):
            q = f"conv4.{b}."
            t = L(q + "conv1", o, q + "conv1", q + "bn1")
            t = L(q + "conv2", t, q + "conv2", q + "bn2", groups=32)
            o = L(q + "conv3", t, q + "conv3", q + "bn3", residual=o)
        pa = L("convPa.0", o, "convPa.0", "convPa.1", stride=2)
        pa = L("convPa.3", pa, "convPa.3", None, relu=False)
        logits = L("convPb", pa, "convPb", None, relu=False)
        da = L("convDa.0", o, "convDa.0", "convDa.1")
        da = L("convDa.3", da, "convDa.3", None, relu=False)
        draw = L("convDb", da, "convDb", None, relu=False)
        sta = L("ConvSta", o, "ConvSta", None, relu=False) if "ConvSta.weight" in self.sd else None
        return logits, draw, sta

    def det(self, x):
        """nets/sfd2.py:313-354: (score [1,1,8H8,8W8], stability [1,1,H,W] or None, desc [1,128,H4,W4] unit norm)."""
        logits, draw, sta = self.det_raw(x)
        semi = torch.exp(logits)
        semi = semi / (semi.sum(dim=1, keepdim=True) + 1e-5)
        semi = semi[:, :-1]
        Hc, Wc = semi.shape[2:]
        score = semi.permute(0, 2, 3, 1).reshape(1, Hc, Wc, 8, 8).permute(0, 1, 3, 2, 4).reshape(1, 1, Hc * 8, Wc * 8)
        desc = F.normalize(draw, dim=1)
        stab = None
        if sta is not None:
            up = F.interpolate(sta, size=x.shape[2:], mode="bilinear", align_corners=False)
            cls = up.argmax(dim=1, keepdim=True)
            stab = torch.tensor([0.1, 0.5, 1.0])[cls]
        return score, stab, desc


def norm_rgb(img):
    m = torch.tensor(RGB_MEAN).view(1, 3, 1, 1)
    s = torch.tensor(RGB_STD).view(1, 3, 1, 1)
    return (img - m) / s


def simple_nms(s, r=4):
    """nets/extractor.py:20-35."""
    def mp(x):
        return F.max_pool2d(x, kernel_size=2 * r + 1, stride=1, padding=r)
    zeros = torch.zeros_like(s)
    mm = s == mp(s)
    for _ in range(2):
        supp = mp(mm.float()) > 0
        ss = torch.where(supp, zeros, s)
        new = ss == mp(ss)
        mm = mm | (new & (~supp))
    return torch.where(mm, s, zeros)


def extract(twin, img, conf_th=0.001, topK=4096):
    """nets/extractor.py:97-338, single scale, no mask.  img [3,H,W] float32 in [0,1].  Tie rule of this repository:
    score descending, then row-major pixel index ascending."""
    x = norm_rgb(_t(img)[None])
    H, W = x.shape[2:]
    with torch.no_grad():
        score, stab, desc = twin.det(x)
        if score.shape[2] != H or score.shape[3] != W:
            score = F.interpolate(score, size=(H, W), mode="bilinear", align_corners=False)
        heat = score * stab if stab is not None else score
        nm = simple_nms(heat, 4)[0, 0]
        ys, xs = torch.nonzero(nm > conf_th, as_tuple=True)
        sc = nm[ys, xs]
        keep = (xs >= 4) & (xs < W - 4) & (ys >= 4) & (ys < H - 4)
        xs, ys, sc = xs[keep], ys[keep], sc[keep]
        order = np.lexsort(((ys * W + xs).numpy(), -sc.numpy()))
        if topK > 0:
            order = order[:topK]
        order = torch.from_numpy(order.copy())
        xs, ys, sc = xs[order], ys[order], sc[order]
        gx = xs.float() / (W / 2.0) - 1.0
        gy = ys.float() / (H / 2.0) - 1.0
        grid = torch.stack([gx, gy], dim=1).view(1, 1, -1, 2)
        d = F.grid_sample(desc, grid, mode="bilinear", align_corners=False)[0, :, 0].t()
        d = d / d.norm(dim=1, keepdim=True)
    return {"keypoints": torch.stack([xs, ys], 1).double().numpy(), "scores": sc.double().numpy(),
            "descriptors": d.double().numpy(), "heat": heat[0, 0].numpy()}


def nnm(d0, d1):
    """hloc NearestNeighbor with do_mutual_check, no thresholds (hloc/matchers/nearest_neighbor.py:6-57).
    d0 [N,128], d1 [M,128] float32."""
    with torch.no_grad():
        sim = _t(d0) @ _t(d1).t()
        s0, m0 = sim.max(dim=1)
        m1 = sim.argmax(dim=0)
        ok = m1[m0] == torch.arange(m0.numel())
        return {"matches0": torch.where(ok, m0, torch.full_like(m0, -1)).numpy(), "matching_scores0": ((s0 + 1) / 2).numpy()}
