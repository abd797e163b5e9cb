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
