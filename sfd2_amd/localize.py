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
