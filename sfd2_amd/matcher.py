"""Mirror of it_loc/matcher.py (reference): Matcher(conf) with conf['model']['name'] in
{'nnm', 'nnr'}; forward takes numpy float64 [N,128] / [M,128] descriptor sets and
returns numpy matches0 / matching_scores0 (it_loc/matcher.py:91-119).  The label path
('nnml', matcher_with_label) is not on the shipped pipelines' path
(it_loc/localize_cv2.py:714 with_label=False) and raises NotImplementedError."""
import ctypes

import numpy as np

from . import _lib


def names_to_pair(name0, name1):  # it_loc/matcher.py:20-21, hloc/utils/parsers.py:66-67
    return '_'.join((name0.replace('/', '-'), name1.replace('/', '-')))


confs = {  # it_loc/matcher.py:24-82 (the entries that do not need external nets)
    'NNM': {'output': 'NNM', 'model': {'name': 'nnm', 'do_mutual_check': True, 'distance_threshold': None}},
    'NNR': {'output': 'NNR', 'model': {'name': 'nnr', 'do_mutual_check': True, 'distance_threshold': 0.9}},
}


class Matcher:
    def __init__(self, conf):
        self.conf = conf
        self.mode = conf['model']['name']
        if self.mode not in ('nnm', 'nnr'):
            raise NotImplementedError(f"matcher mode {self.mode!r}: only 'nnm' and 'nnr' are on the hot path")
        self.sim_mode = conf['model'].get('sim_mode', 'f16')
        self._device = 0

    def eval(self):
        return self

    def cuda(self, device=None):
        if isinstance(device, int):
            self._device = device
        return self

    def _conf(self):
        flavour = _lib.MATCH_ITLOC_NNM if self.mode == 'nnm' else _lib.MATCH_ITLOC_NNR
        ratio = float(self.conf['model'].get('distance_threshold') or 0.0)
        return _lib.MatchConf(flavour, 1, ratio, 0.0, _lib.SIM_F16X2 if self.sim_mode == 'f16x2' else _lib.SIM_F16)

    def forward(self, data):
        d0 = np.ascontiguousarray(data['descriptors0'])
        d1 = np.ascontiguousarray(data['descriptors1'])
        dt = {np.dtype(np.float64): _lib.DT_F64, np.dtype(np.float32): _lib.DT_F32, np.dtype(np.float16): _lib.DT_F16}
        if d0.dtype not in dt or d1.dtype != d0.dtype:
            d0, d1 = d0.astype(np.float64), d1.astype(np.float64)
        n0, dim = d0.shape
        n1 = d1.shape[0]
        ctx = _lib.default_context(self._device)
        m = np.empty((n0,), dtype=np.int64)
        s = np.empty((n0,), dtype=np.float32)
        conf = self._conf()
        _lib.check(ctx.lib.sfd2_match(ctx.h, d0.ctypes.data, n0, d1.ctypes.data, n1, dim, dt[d0.dtype], _lib.LAYOUT_ND,
                                      0, ctypes.byref(conf), m.ctypes.data, s.ctypes.data, 0))
        return {'matches0': m, 'matching_scores0': s.astype(np.float64)}

    __call__ = forward

    def match_batch(self, d0, d1_list, rows_list=None):
        """One query against K database descriptor sets in a single launch (the localiser's
        inner loop, it_loc/localize_cv2.py:705-715).  Returns ([K,N] matches0, [K,N] scores0).
        rows_list[i] (int array or None) restricts database set i to those rows; matches0 then
        reports the caller's row indices (feature_matching's mask + remap, :531-534,557-559)."""
        d0 = np.ascontiguousarray(d0, dtype=np.float32)
        ds = [np.ascontiguousarray(d, dtype=np.float32) for d in d1_list]
        k, n0 = len(ds), d0.shape[0]
        rows = [None] * k if rows_list is None else [None if r is None else np.ascontiguousarray(r, dtype=np.int32)
                                                     for r in rows_list]
        ctx = _lib.default_context(self._device)
        q = _lib.DescSet(d0.ctypes.data, n0, _lib.DT_F32, _lib.LAYOUT_ND, 0, None, 0, 0)
        db = (_lib.DescSet * k)(*[_lib.DescSet(d.ctypes.data, d.shape[0], _lib.DT_F32, _lib.LAYOUT_ND, 0,
                                               None if r is None else r.ctypes.data, 0 if r is None else len(r), 0)
                                  for d, r in zip(ds, rows)])
        m = np.empty((k, n0), dtype=np.int64)
        s = np.empty((k, n0), dtype=np.float32)
        conf = self._conf()
        _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(q), db, k, d0.shape[1], ctypes.byref(conf),
                                            m.ctypes.data, s.ctypes.data, 0, 0))
        return m, s
