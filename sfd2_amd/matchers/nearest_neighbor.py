"""Mirror of hloc/matchers/nearest_neighbor.py (reference): the NearestNeighbor plugin,
discoverable by dynamic_load as the only BaseModel subclass of this module.
The einsum + top-k + mutual check (nearest_neighbor.py:38-57) run as one fused
MFMA GEMM + top-2 kernel in libsfd2hip (sfd2_match); the similarity matrix is
never materialised."""
import numpy as np

from .. import _lib
from ..base_model import BaseModel

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


class NearestNeighbor(BaseModel):
    default_conf = {
        'ratio_threshold': None,
        'distance_threshold': None,
        'do_mutual_check': True,
        'sim_mode': 'f16',   # 'f16' (default) or 'f16x2' (hi+lo split, ~fp32 accuracy)
    }
    required_data_keys = ['descriptors0', 'descriptors1']
    required_inputs = ['descriptors0', 'descriptors1']

    def _init(self, conf):
        pass

    def match_conf(self):
        """The C-ABI form of this plugin's conf (sfd2_match_conf)."""
        return _lib.MatchConf(_lib.MATCH_HLOC, int(bool(self.conf['do_mutual_check'])),
                              float(self.conf['ratio_threshold'] or 0.0), float(self.conf['distance_threshold'] or 0.0),
                              _lib.SIM_F16X2 if self.conf['sim_mode'] == 'f16x2' else _lib.SIM_F16)

    def _forward(self, data):
        d0, d1 = data['descriptors0'], data['descriptors1']   # [B, D, N], [B, D, M]
        is_t = torch is not None and isinstance(d0, torch.Tensor)
        on_dev = bool(is_t and d0.is_cuda)
        if is_t:
            a0 = d0.detach().to(torch.float32).contiguous()
            a1 = d1.detach().to(torch.float32).contiguous()
            if on_dev:
                torch.cuda.current_stream(a0.device).synchronize()
        else:
            a0 = np.ascontiguousarray(d0, dtype=np.float32)
            a1 = np.ascontiguousarray(d1, dtype=np.float32)
        B, D, N = a0.shape
        M = a1.shape[2]
        dev = a0.device.index if on_dev and a0.device.index is not None else 0
        ctx = _lib.default_context(dev)
        conf = self.match_conf()
        if on_dev:
            m = torch.empty((B, N), dtype=torch.int64, device=a0.device)
            s = torch.empty((B, N), dtype=torch.float32, device=a0.device)
        else:
            m = np.empty((B, N), dtype=np.int64)
            s = np.empty((B, N), dtype=np.float32)
        for b in range(B):
            _lib.check(ctx.lib.sfd2_match(ctx.h, _lib.ptr(a0[b]), N, _lib.ptr(a1[b]), M, D, _lib.DT_F32, _lib.LAYOUT_DN,
                                          int(on_dev), conf, _lib.ptr(m[b]), _lib.ptr(s[b]), int(on_dev)))
        if is_t and not on_dev:
            m, s = torch.from_numpy(m), torch.from_numpy(s)
        return {'matches0': m, 'matching_scores0': s}
