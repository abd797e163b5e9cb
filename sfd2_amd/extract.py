"""Host-side mirror of extract.py (reference), the older SuperPoint-style extractor:
nms_fast (:17-84), extrat_spp_feats_multiscale (:87-201), extract_spp_feats_singlescale (:205-277),
extract_spp_return (:280-302).  Not used by the shipped pipelines but on the hot-path table
(8a18, 8f rank 3).  The greedy grid NMS runs on the GPU as an exact parallel relaxation (libsfd2hip sfd2_nms_fast /
sfd2_extract_spp / sfd2_extract_spp_levels)."""
import ctypes

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def nms_fast(in_corners, H, W, dist_thresh):
    """extract.py:17-84.  in_corners: 3xN [x, y, conf] with integer pixel coordinates (every caller
    passes np.where() output).  Returns (out 3xK sorted by confidence descending, out_inds K)."""
    in_corners = np.asarray(in_corners, dtype=np.float64)
    n = in_corners.shape[1]
    if n == 0:
        return np.zeros((3, 0)).astype(int), np.zeros(0).astype(int)
    xs = np.rint(in_corners[0]).astype(np.int64)
    ys = np.rint(in_corners[1]).astype(np.int64)
    if n == 1:
        out = np.vstack((xs, ys, in_corners[2])).reshape(3, 1)
        return out, np.zeros((1)).astype(int)
    sc = in_corners[2].astype(np.float32)
    heat = np.zeros((H, W), dtype=np.float32)
    heat[ys, xs] = sc
    if (sc <= 0).any() or len(np.unique(ys * W + xs)) != n:
        raise ValueError("nms_fast expects positive scores at distinct integer pixels")
    kept = np.empty_like(heat)
    ctx = _lib.default_context(0)
    th = float(np.nextafter(np.float32(sc.min()), np.float32(0)))   # every given corner is a candidate
    _lib.check(ctx.lib.sfd2_nms_fast(ctx.h, heat.ctypes.data, H, W, th, int(dist_thresh), kept.ctypes.data))
    idx_of = {(int(x), int(y)): i for i, (x, y) in enumerate(zip(xs, ys))}
    ky, kx = np.nonzero(kept)
    inds = np.array([idx_of[(int(x), int(y))] for x, y in zip(kx, ky)], dtype=np.int64)
    order = np.lexsort((ky * W + kx, -kept[ky, kx]))           # score descending, pixel index ascending
    inds = inds[order]
    return in_corners[:, inds], inds


def extract_spp_feats_singlescale(model, img, conf_th=0.10):
    """extract.py:205-277.  img: [1,3,H,W] NORMALISED image (the caller applies norm_RGB).
    Returns (pts [N,3] f64 (x, y, score), desc [N,128] f32, scores [N] f64,
             desc_full [128,Hc,Wc] f32, heatmap [H,W] f32)."""
    ctx = model.context
    if torch is not None and isinstance(img, torch.Tensor):
        a = img.detach().to(torch.float32).cpu().numpy()
    else:
        a = np.asarray(img, dtype=np.float32)
    a = np.ascontiguousarray(a.reshape(a.shape[-3:]))
    _, H, W = a.shape
    h2, w2 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hc, wc = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    cap = H * W
    kp = np.empty((cap, 2), dtype=np.float32)
    sc = np.empty((cap,), dtype=np.float32)
    de = np.empty((cap, 128), dtype=np.float32)
    heat = np.empty((H, W), dtype=np.float32)
    dfull = np.empty((128, hc, wc), dtype=np.float32)
    n = ctypes.c_int(0)
    flags = 0 if getattr(model, "require_stability", True) else _lib.FLAG_NO_STABILITY
    _lib.check(ctx.lib.sfd2_extract_spp(ctx.h, a.ctypes.data, 0, H, W, float(conf_th), flags, kp.ctypes.data,
                                        sc.ctypes.data, de.ctypes.data, cap, ctypes.byref(n), heat.ctypes.data,
                                        dfull.ctypes.data))
    n = n.value
    pts = np.concatenate([kp[:n].astype(np.float64), sc[:n, None].astype(np.float64)], axis=1)
    return pts, de[:n].copy(), pts[:, 2].copy(), dfull, heat


def spp_level_schedule(H, W, scale_f=2 ** 0.25, min_scale=0.05, max_scale=1.0, min_size=256, max_size=2048):
    """The while-loop of extract.py:102-188 as a list of (nh, nw, emit): Python float arithmetic and round() exactly as
    the reference (s /= scale_f; nh, nw = round(H * s), round(W * s))."""
    assert max_scale <= 1
    s = 1.0
    nh, nw = H, W
    levels = []
    while s + 0.001 >= max(min_scale, min_size / max(H, W)):
        levels.append((nh, nw, s - 0.001 <= min(max_scale, max_size / max(H, W))))
        s /= scale_f
        nh, nw = round(H * s), round(W * s)
    return levels


def extrat_spp_feats_multiscale(model, img, conf_th=0.0050, scale_f=2 ** 0.25,
                                min_scale=0.05, max_scale=1.0, min_size=256, max_size=2048):
    """extract.py:87-201 (the reference's spelling).  img: [1,3,H,W] NORMALISED image.  Returns
    (all_pts [N,3] f64 (x, y, score) in original-image coordinates, all_descs [N,128], all_scores [N]) with the levels
    concatenated in scale order, or (None, None, None) when no level is emitted."""
    ctx = model.context
    if torch is not None and isinstance(img, torch.Tensor):
        a = img.detach().to(torch.float32).cpu().numpy()
    else:
        a = np.asarray(img, dtype=np.float32)
    a = np.ascontiguousarray(a.reshape(a.shape[-3:]))
    _, H, W = a.shape
    levels = spp_level_schedule(H, W, scale_f, min_scale, max_scale, min_size, max_size)
    if not any(e for _, _, e in levels):
        return None, None, None
    nl = len(levels)
    nh = (ctypes.c_int32 * nl)(*[l[0] for l in levels])
    nw = (ctypes.c_int32 * nl)(*[l[1] for l in levels])
    em = (ctypes.c_int32 * nl)(*[int(l[2]) for l in levels])
    # Output capacity: the greedy grid NMS (extract.py:17-84, dist 4) keeps at most one point per 5 x 5 pixels,
    # so one slot per 4 x 4 cell of every emitted level always suffices -- 1/16 of the worst case "every pixel" (6.3 M rows,
    # 3.2 GB of descriptors, for one 1600x1200 pyramid).  The library reports "capacity exceeded" should that ever be wrong;
    # the call is then repeated once with the worst case.
    cap = sum(((l[0] + 3) // 4) * ((l[1] + 3) // 4) for l in levels if l[2])
    for attempt in range(2):
        kp = np.empty((cap, 2), dtype=np.float32)
        sc = np.empty((cap,), dtype=np.float32)
        de = np.empty((cap, 128), dtype=np.float32)
        cnt = (ctypes.c_int32 * nl)()
        rc = ctx.lib.sfd2_extract_spp_levels(ctx.h, a.ctypes.data, 0, H, W, nl, nh, nw, em, float(conf_th), 0,
                                             kp.ctypes.data, sc.ctypes.data, de.ctypes.data, cap, cnt)
        if rc != 0 and attempt == 0 and b"capacity exceeded" in (ctx.lib.sfd2_last_error() or b""):
            cap = sum(l[0] * l[1] for l in levels if l[2])
            continue
        _lib.check(rc)
        break
    all_pts, all_descs = [], []
    off = 0
    for (lh, lw, emit), n in zip(levels, cnt):
        if not emit:
            continue
        pts = np.zeros((n, 3))                                   # float64, as the reference's np.zeros((3, N))
        pts[:, 0] = kp[off:off + n, 0].astype(np.float64) * W / lw     # extract.py:176-177
        pts[:, 1] = kp[off:off + n, 1].astype(np.float64) * H / lh
        pts[:, 2] = sc[off:off + n]
        all_pts.append(pts)
        all_descs.append(de[off:off + n].copy())
        off += n
    all_pts = np.vstack(all_pts)
    return all_pts, np.vstack(all_descs), all_pts[:, 2]


def extract_spp_return(sgd2, img_path, conf_th=0.1, need_nms=False, multi_scale=False, min_size=256, max_size=9999):
    """extract.py:280-302 (tensor input; a str path needs the caller's own image decoding)."""
    if isinstance(img_path, str):
        raise NotImplementedError("pass the normalised image tensor [1,3,H,W]; image decoding is outside the hot path")
    if multi_scale:      # extract.py:295-299: scale_f = 1.2
        return extrat_spp_feats_multiscale(model=sgd2, img=img_path, conf_th=conf_th, scale_f=1.2, min_size=min_size,
                                           max_size=max_size)
    return extract_spp_feats_singlescale(model=sgd2, img=img_path, conf_th=conf_th)
