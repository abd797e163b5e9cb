"""Caller-side mirror of it_loc/localize_cv2.py:511-560 feature_matching: mask the database
descriptors to those with a triangulated 3D point, early-out when 3 or fewer remain, call the
matcher, map the match indices back to the unmasked database indexing."""
import numpy as np


def feature_matching(desc_q, desc_db, matcher, label_q=None, label_db=None, db_3D_ids=None):
    with_label = (label_q is not None and label_db is not None)
    if with_label:
        raise NotImplementedError("label-aware matching is not on the shipped pipelines' path "
                                  "(it_loc/localize_cv2.py:714 with_label=False)")
    if db_3D_ids is None:
        return matcher({"descriptors0": desc_q, "descriptors1": desc_db})["matches0"]
    masks = (np.asarray(db_3D_ids) != -1)
    if np.sum(masks) <= 3:
        return np.ones((desc_q.shape[0],), dtype=int) * -1
    valid_ids = np.flatnonzero(masks)
    matches = np.array(matcher({"descriptors0": desc_q, "descriptors1": desc_db[masks]})["matches0"])
    hit = matches >= 0
    matches[hit] = valid_ids[matches[hit]]
    return matches


def feature_matching_batch(desc_q, desc_db_list, matcher, db_3D_ids_list):
    """The loop of match_cluster_2D (it_loc/localize_cv2.py:563-590) over one query's retrieved
    database images as ONE device call: per image the 3D-point mask, the <= 3 early-out
    (:536-537), mutual-NN matching and the index remap all happen inside sfd2_match_batch
    (row-selected database sets), so nothing is gathered or remapped on the host.
    matcher: sfd2_amd.matcher.Matcher.  Returns a list of [N] int arrays (matches per image)."""
    k = len(desc_db_list)
    out = [None] * k
    live, rows = [], []
    for i, (d, ids) in enumerate(zip(desc_db_list, db_3D_ids_list)):
        if ids is None:
            live.append(i); rows.append(None)
            continue
        valid = np.flatnonzero(np.asarray(ids) != -1).astype(np.int32)
        if len(valid) <= 3:
            out[i] = np.ones((desc_q.shape[0],), dtype=int) * -1
        else:
            live.append(i); rows.append(valid)
    if live:
        m, _ = matcher.match_batch(desc_q, [desc_db_list[i] for i in live], rows)
        for j, i in enumerate(live):
            out[i] = m[j].astype(int)
    return out


class StoreMatcher:
    """The localiser's matching loop against a FEATURE STORE with the database descriptor sets resident in HBM (round 5).

    it_loc/localize_cv2.py:563-590 reads, for every query and every retrieved database image, that image's descriptors from the
    feature file (:571-574), masks them to the key points with a 3D point, uploads both sets and runs one matcher call.
    Here a database image's set is read and converted ONCE (sfd2_desc_pack, through sfd2_amd.pipeline.ResidentSets: LRU in HBM,
    fp16 [n][128]) and a query costs one sfd2_match_batch whose row selections (the 3D-point masks) are the only per-query
    uploads.  Results equal feature_matching_batch on the same arrays (same conversion kernel, same matcher kernels).

    matcher: sfd2_amd.matcher.Matcher (mode nnm / nnr); feats: an open feature store (sfd2_amd.feature_io.open_store)."""

    def __init__(self, matcher, feats, cache_bytes=None, readers=4):
        from . import _lib
        from .pipeline import ResidentSets
        self.matcher = matcher
        self.ctx = _lib.default_context(matcher._device)
        self.sets = ResidentSets(self.ctx, feats, budget=cache_bytes, readers=readers)
        self._seq = 0

    def prefetch(self, db_names):
        """Start reading sets a coming query will need (host side; the conversion happens at first use)."""
        self.sets.prefetch(db_names)

    def match(self, desc_q, db_names, db_3D_ids_list=None):
        """desc_q: [N,128] array (float64 / float32) or the NAME of a set in the store; db_names: the retrieved database images;
        db_3D_ids_list[i]: that image's point3D ids (-1 = none) or None.  Returns a list of [N] int arrays (matches0 per image,
        indices into the UNMASKED database key points, -1 = no match) -- feature_matching's contract."""
        import ctypes
        from . import _lib
        k = len(db_names)
        ids_list = [None] * k if db_3D_ids_list is None else list(db_3D_ids_list)
        seq = self._seq
        self._seq += 1
        self.sets.completed_seq = seq - 1          # match() is synchronous: every earlier query has finished
        if isinstance(desc_q, str):
            qp, n0 = self.sets.get(desc_q, seq)
            q = _lib.DescSet(qp, n0, _lib.DT_F16, _lib.LAYOUT_ND, 1, None, 0, 0)
            keep_q = None
        else:
            keep_q = np.ascontiguousarray(desc_q)
            if keep_q.dtype not in (np.float64, np.float32):
                keep_q = keep_q.astype(np.float64)
            n0 = keep_q.shape[0]
            q = _lib.DescSet(keep_q.ctypes.data, n0, _lib.DT_F64 if keep_q.dtype == np.float64 else _lib.DT_F32, _lib.LAYOUT_ND, 0, None, 0, 0)
        out = [None] * k
        live, rows, sets = [], [], []
        for i, (name, ids) in enumerate(zip(db_names, ids_list)):
            r = None
            if ids is not None:
                r = np.flatnonzero(np.asarray(ids) != -1).astype(np.int32)
                if len(r) <= 3:                       # localize_cv2.py:536-537
                    out[i] = np.ones((n0,), dtype=int) * -1
                    continue
            p, n1 = self.sets.get(name, seq)
            if ids is not None and len(np.asarray(ids)) != n1:
                raise ValueError(f"{name}: {len(np.asarray(ids))} point ids for {n1} key points")
            live.append(i); rows.append(r); sets.append((p, n1))
        if live and n0 > 0:
            db = (_lib.DescSet * len(live))(*[_lib.DescSet(p, n1, _lib.DT_F16, _lib.LAYOUT_ND, 1, None if r is None else r.ctypes.data,
                                                           0 if r is None else len(r), 0) for (p, n1), r in zip(sets, rows)])
            m = np.empty((len(live), n0), dtype=np.int64)
            s = np.empty((len(live), n0), dtype=np.float32)
            conf = self.matcher._conf()
            _lib.check(self.ctx.lib.sfd2_match_batch(self.ctx.h, ctypes.byref(q), db, len(live), 128, ctypes.byref(conf), m.ctypes.data,
                                                     s.ctypes.data, 0, 0))
            for j, i in enumerate(live):
                out[i] = m[j].astype(int)
        else:
            for i in live:
                out[i] = np.ones((n0,), dtype=int) * -1
        return out

    def close(self):
        self.sets.close()
