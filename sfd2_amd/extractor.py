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


def extract_resnet_return(model, img, conf_th=0.001, mask=None, topK=-1, **kwargs):
    """nets/extractor.py:97-338.  img: [1,3,H,W] in [0,1] (RGB), cpu or cuda.
    Returns {'keypoints': [N,2] f64 (x,y), 'descriptors': [N,128] f64, 'scores': [N] f64}
    sorted by score descending, N <= topK (topK <= 0: all candidates)."""
    if mask is not None:
        raise NotImplementedError("semantic-mask top-K branch (nets/extractor.py:240-319) is not on the shipped "
                                  "pipelines' path (extract_localization.py:247 passes mask=None)")
    scales = kwargs.get("scales", [1.0])
    if list(scales) != [1.0]:
        raise NotImplementedError("multi-scale extraction (nets/extractor.py:118-124); every shipped conf uses [1.0]")
    ctx = model.context
    arr, on_dev, H, W = _to_chw(img)
    flags = 0 if getattr(model, "require_stability", True) else _lib.FLAG_NO_STABILITY
    cap = int(topK) if topK > 0 else max(65536, (H * W) // 8)
    kp = np.empty((cap, 2), dtype=np.float32)
    sc = np.empty((cap,), dtype=np.float32)
    de = np.empty((cap, 128), dtype=np.float32)
    n = ctypes.c_int(0)
    _lib.check(ctx.lib.sfd2_extract(ctx.h, _lib.ptr(arr), int(on_dev), H, W, float(conf_th), int(topK), flags,
                                    kp.ctypes.data, sc.ctypes.data, de.ctypes.data, 0, cap, ctypes.byref(n)))
    n = n.value
    # the reference returns float64 containers (nets/extractor.py:322-337)
    return {"keypoints": kp[:n].astype(np.float64), "descriptors": de[:n].astype(np.float64),
            "scores": sc[:n].astype(np.float64)}
