"""Mirror of it_loc/matcher.py (reference): Matcher(conf) with conf['model']['name'] in
{'nnm', 'nnr', 'nnml'}; forward takes numpy float64 [N,128] / [M,128] descriptor sets and
returns numpy matches0 / matching_scores0 (it_loc/matcher.py:91-119).  'nnml' is the label-aware
matcher (matcher_with_label, :239-297; disabled in the shipped pipelines, with_label=False at
it_loc/localize_cv2.py:714): mutual NN inside every label the two images share, then mutual NN
among the key points that are still unmatched; every similarity / arg-max runs on the device."""
import ctypes

import numpy as np

from . import _lib


def names_to_pair(name0, name1):  # it_loc/matcher.py:20-21, hloc/utils/parsers.py:66-67
    return '_'.join((name0.replace('/', '-'), name1.replace('/', '-')))


confs = {  # it_loc/matcher.py:24-82 (the entries that do not need external nets)
    'NNML': {'output': 'NNML', 'model': {'name': 'nnml', 'do_mutual_check': True, 'distance_threshold': None}},
    'NNM': {'output': 'NNM', 'model': {'name': 'nnm', 'do_mutual_check': True, 'distance_threshold': None}},
    'NNR': {'output': 'NNR', 'model': {'name': 'nnr', 'do_mutual_check': True, 'distance_threshold': 0.9}},
}


class Matcher:
    def __init__(self, conf):
        self.conf = conf
        self.mode = conf['model']['name']
        if self.mode not in ('nnm', 'nnr', 'nnml'):
            raise NotImplementedError(f"matcher mode {self.mode!r}: 'nnm', 'nnr' and 'nnml' are implemented")
        self.sim_mode = conf['model'].get('sim_mode', 'f16')
        self._device = 0

    def eval(self):
        return self

    def cuda(self, device=None):
        if isinstance(device, int):
            self._device = device
        return self

    def _conf(self):
        flavour = _lib.MATCH_ITLOC_NNR if self.mode == 'nnr' else _lib.MATCH_ITLOC_NNM
        ratio = float(self.conf['model'].get('distance_threshold') or 0.0)
        return _lib.MatchConf(flavour, 1, ratio, 0.0, _lib.SIM_F16X2 if self.sim_mode == 'f16x2' else _lib.SIM_F16)

    def _match(self, d0, d1):
        """One device call: (matches0 [n0] int64, top-1 similarity [n0] f32) in the mode's flavour."""
        d0 = np.ascontiguousarray(d0)
        d1 = np.ascontiguousarray(d1)
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
        return m, s

    def _match_segments(self, d0, d1, seg0, seg1):
        """One device call over all segments: rows [seg0[s], seg0[s+1]) of d0 against rows [seg1[s], seg1[s+1]) of d1."""
        d0 = np.ascontiguousarray(d0, dtype=np.float64)
        d1 = np.ascontiguousarray(d1, dtype=np.float64)
        seg0 = np.ascontiguousarray(seg0, dtype=np.int32)
        seg1 = np.ascontiguousarray(seg1, dtype=np.int32)
        ctx = _lib.default_context(self._device)
        m = np.empty((d0.shape[0],), dtype=np.int64)
        s = np.empty((d0.shape[0],), dtype=np.float32)
        conf = self._conf()
        _lib.check(ctx.lib.sfd2_match_segments(ctx.h, d0.ctypes.data, d0.shape[0], d1.ctypes.data, d1.shape[0], d0.shape[1],
                                               _lib.DT_F64, _lib.LAYOUT_ND, 0, len(seg0) - 1, seg0.ctypes.data, seg1.ctypes.data,
                                               ctypes.byref(conf), m.ctypes.data, s.ctypes.data, 0))
        return m, s

    def matcher_with_label(self, descriptors1, labels1, descriptors2, labels2):
        """it_loc/matcher.py:239-297.  Returns [M, 2] (index in 1, index in 2), in the reference's order: same-label
        matches label by label (ascending label, ascending index in 1), then the matches among the rest.
        Device work: ONE segmented launch for every shared label (both sets ordered by label, the diagonal blocks of the
        block-masked similarity matrix), one launch for the rest; host work is vectorised numpy."""
        d1 = np.asarray(descriptors1)
        d2 = np.asarray(descriptors2)
        labels1, labels2 = np.asarray(labels1).reshape(-1), np.asarray(labels2).reshape(-1)
        shared = np.intersect1d(np.unique(labels1), np.unique(labels2))
        shared = shared[shared > 0]
        all_matches = np.zeros((0, 2), dtype=int)
        if len(shared) and len(d1) and len(d2):
            # rows that carry a shared label, grouped by label (stable: ascending original index inside a label)
            k1 = np.flatnonzero(np.isin(labels1, shared))
            k2 = np.flatnonzero(np.isin(labels2, shared))
            o1 = k1[np.argsort(labels1[k1], kind="stable")]
            o2 = k2[np.argsort(labels2[k2], kind="stable")]
            seg0 = np.concatenate([[0], np.cumsum([np.count_nonzero(labels1[o1] == u) for u in shared])])
            seg1 = np.concatenate([[0], np.cumsum([np.count_nonzero(labels2[o2] == u) for u in shared])])
            m, _ = self._match_segments(d1[o1], d2[o2], seg0, seg1)
            hit = np.flatnonzero(m >= 0)                              # label-major, then ascending row: the reference's order
            all_matches = np.stack([o1[hit], o2[m[hit]]], axis=1).astype(int)
        rest1 = np.setdiff1d(np.arange(len(d1)), all_matches[:, 0])   # (:266-281) everything still unmatched, any label
        rest2 = np.setdiff1d(np.arange(len(d2)), all_matches[:, 1])
        if len(rest1) and len(rest2):                                 # mutual NN among the rest (:283-293)
            m, _ = self._match(d1[rest1], d2[rest2])
            hit = np.flatnonzero(m >= 0)
            all_matches = np.concatenate([all_matches, np.stack([rest1[hit], rest2[m[hit]]], axis=1).astype(int)], axis=0)
        return all_matches.reshape(-1, 2)

    def forward(self, data):
        if self.mode == 'nnml':
            d0, d1 = np.asarray(data['descriptors0']), np.asarray(data['descriptors1'])
            pairs = self.matcher_with_label(d0, data['labels0'], d1, data['labels1'])
            all_matches = np.ones((d0.shape[0],), dtype=int) * -1
            all_matches[pairs[:, 0]] = pairs[:, 1]
            _, s = self._match(d0, d1)                           # scores: top-1 similarity over ALL of image 2 (:111)
            return {'matches0': all_matches, 'matching_scores0': s.astype(np.float64)}
        m, s = self._match(data['descriptors0'], data['descriptors1'])
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
