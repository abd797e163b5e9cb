"""Host-side mirror of nets/extractor.py (reference): same function names, argument
meaning, dict keys, dtypes and error behaviour on the single-scale, mask-free
pipeline path (extract_localization.py:245-250).  All arithmetic runs in
libsfd2hip on the MI355X; nothing here computes on the CPU.
"""
import ctypes

import numpy as np

from . import _lib

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

RGB_mean = [0.485, 0.456, 0.406]  # nets/extractor.py:14
RGB_std = [0.229, 0.224, 0.225]   # nets/extractor.py:15


def _to_chw(img):
    """[1,3,H,W] / [3,H,W] torch or numpy -> (contiguous fp32 array-like, on_device, H, W)."""
    if torch is not None and isinstance(img, torch.Tensor):
        t = img.detach().to(torch.float32)
        t = t.reshape(t.shape[-3:]).contiguous()
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
            return t, True, int(t.shape[1]), int(t.shape[2])
        a = t.numpy()
        return a, False, a.shape[1], a.shape[2]
    a = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
    a = a.reshape(a.shape[-3:])
    return a, False, a.shape[1], a.shape[2]


def simple_nms(scores, nms_radius: int):
    """nets/extractor.py:20-35.  scores: [1,1,H,W] or [H,W]; returns same shape/type."""
    assert nms_radius >= 0
    is_t = torch is not None and isinstance(scores, torch.Tensor)
    a = scores.detach().cpu().numpy() if is_t else np.asarray(scores)
    shp = a.shape
    m = np.ascontiguousarray(a.reshape(shp[-2:]), dtype=np.float32)
    out = np.empty_like(m)
    ctx = _lib.default_context(0)
    _lib.check(ctx.lib.sfd2_simple_nms(ctx.h, m.ctypes.data, m.shape[0], m.shape[1], int(nms_radius), out.ctypes.data))
    out = out.reshape(shp)
    return torch.from_numpy(out).to(scores.device) if is_t else out


def _to_u8_hwc(img):
    """uint8 [H,W,3] (or [1,H,W,3]) torch / numpy -> (contiguous array-like, on_device, H, W), else None."""
    if torch is not None and isinstance(img, torch.Tensor):
        if img.dtype != torch.uint8:
            return None
        t = img.detach().reshape(img.shape[-3:]).contiguous()
        if t.shape[-1] != 3:
            raise ValueError("uint8 images must be [H, W, 3] (HWC)")
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()
            return t, True, int(t.shape[0]), int(t.shape[1])
        a = t.numpy()
        return a, False, a.shape[0], a.shape[1]
    a = np.asarray(img)
    if a.dtype != np.uint8:
        return None
    a = np.ascontiguousarray(a.reshape(a.shape[-3:]))
    if a.shape[-1] != 3:
        raise ValueError("uint8 images must be [H, W, 3] (HWC)")
    return a, False, a.shape[0], a.shape[1]


def select_with_mask(keypoints, scores, descriptors, mask, topK):
    """nets/extractor.py:240-319 on the score-sorted candidate list: label = B + 256 G + 65536 R of the mask pixel
    under the key point (mask is the cv2 BGR image, :252); labelled key points (label != 0) are kept first, the
    unlabelled ones only fill what is left of topK.  Host logic, exactly as in the reference (including its
    behaviour for topK <= 0: the full lists with the labels of the labelled points only)."""
    mask = np.asarray(mask)
    id_img = np.int32(mask[:, :, 2]) * 256 * 256 + np.int32(mask[:, :, 1]) * 256 + np.int32(mask[:, :, 0])
    gid = id_img[keypoints[:, 1].astype(np.int64), keypoints[:, 0].astype(np.int64)] if len(keypoints) else np.zeros((0,), np.int32)
    w = np.flatnonzero(gid != 0)
    wo = np.flatnonzero(gid == 0)
    labels = gid[w].tolist()
    if topK > 0:
        if topK <= len(w):
            idx = np.array(scores[w], dtype=float).argsort()[::-1][:topK]
            sel = w[idx]
            labels = np.array(labels, np.int32)[idx]
        elif topK >= len(w) + len(wo):
            sel = np.concatenate([w, wo])
            labels = labels + [0] * len(wo)
        else:
            idx = np.array(scores[wo], dtype=float).argsort()[::-1][:topK - len(w)]
            sel = np.concatenate([w, wo[idx]])
            labels = labels + [0] * len(idx)
        keypoints, scores, descriptors = keypoints[sel], scores[sel], descriptors[sel]
    return {"keypoints": np.array(keypoints, dtype=float), "descriptors": np.array(descriptors, dtype=float),
            "scores": np.array(scores, dtype=float), "labels": np.array(labels, np.int32)}


def extract_resnet_return(model, img, conf_th=0.001, mask=None, topK=-1, **kwargs):
    """nets/extractor.py:97-338.  img: [1,3,H,W] in [0,1] (RGB), cpu or cuda -- or the decoder's
    uint8 [H,W,3] image as is (kwarg bgr=True for cv2.imread order): the astype(float32) / 255.
    of extract_localization.py:168,186 then happens on the device and only a quarter of the bytes
    cross PCIe.  kwarg scales (default [1.0]) is the reference's pyramid (:113-124).
    Returns {'keypoints': [N,2] f64 (x,y), 'descriptors': [N,128] f64, 'scores': [N] f64}
    sorted by score descending, N <= topK (topK <= 0: all candidates; with several scales the
    reference then returns the per-scale lists concatenated, and so does this)."""
    if mask is not None:
        # semantic-mask branch (nets/extractor.py:240-319): needs EVERY candidate, labelled ones are preferred
        allp = extract_resnet_return(model, img, conf_th=conf_th, mask=None, topK=-1, **kwargs)
        return select_with_mask(allp["keypoints"], allp["scores"], allp["descriptors"], mask, topK)
    scales = [float(s) for s in kwargs.get("scales", [1.0])]
    ctx = model.context
    flags = 0 if getattr(model, "require_stability", True) else _lib.FLAG_NO_STABILITY
    u8 = _to_u8_hwc(img)
    if u8 is not None:
        arr, on_dev, H, W = u8
        flags |= _lib.FLAG_IMG_U8_HWC | (_lib.FLAG_IMG_BGR if kwargs.get("bgr", False) else 0)
    else:
        if kwargs.get("bgr", False):
            raise ValueError("bgr=True needs a uint8 HWC image")
        arr, on_dev, H, W = _to_chw(img)
    n = ctypes.c_int(0)
    if scales == [1.0]:
        cap = int(topK) if topK > 0 else max(65536, (H * W) // 8)
        kp = np.empty((cap, 2), dtype=np.float32)
        sc = np.empty((cap,), dtype=np.float32)
        de = np.empty((cap, 128), dtype=np.float32)
        _lib.check(ctx.lib.sfd2_extract(ctx.h, _lib.ptr(arr), int(on_dev), H, W, float(conf_th), int(topK), flags,
                                        kp.ctypes.data, sc.ctypes.data, de.ctypes.data, 0, cap, ctypes.byref(n)))
    else:
        if not 1 <= len(scales) <= 8:
            raise ValueError("1..8 scales")
        if topK > 0:
            cap = int(topK)
        else:
            cap = sum(max(65536, (int(H * s) * int(W * s)) // 8) for s in scales)
        kp = np.empty((cap, 2), dtype=np.float32)
        sc = np.empty((cap,), dtype=np.float32)
        de = np.empty((cap, 128), dtype=np.float32)
        sarr = (ctypes.c_double * len(scales))(*scales)
        _lib.check(ctx.lib.sfd2_extract_multiscale(ctx.h, _lib.ptr(arr), int(on_dev), H, W, sarr, len(scales),
                                                   float(conf_th), int(topK), flags, kp.ctypes.data, sc.ctypes.data,
                                                   de.ctypes.data, 0, cap, ctypes.byref(n)))
    n = n.value
    # the reference returns float64 containers (nets/extractor.py:322-337)
    return {"keypoints": kp[:n].astype(np.float64), "descriptors": de[:n].astype(np.float64),
            "scores": sc[:n].astype(np.float64)}
