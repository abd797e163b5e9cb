"""Host driver pieces mirroring hloc/match_features.py (reference): confs (:20-45), pair
naming and de-duplication (:87-97), the per-pair call (:99-119) with the int16 / fp16 casts
of the stored results.  HDF5 I/O (h5py) is outside the hot path."""
import os

import numpy as np

from . import matchers
from .base_model import dynamic_load

confs = {
    'NNM': {'output': 'NNM', 'model': {'name': 'nearest_neighbor', 'do_mutual_check': True, 'distance_threshold': None}},
    'ONN': {'output': 'ONN', 'model': {'name': 'nearest_neighbor', 'do_mutual_check': False, 'distance_threshold': None}},
    'NNR': {'output': 'NNR', 'model': {'name': 'nearest_neighbor', 'do_mutual_check': True, 'distance_threshold': 0.9}},
}


def names_to_pair(name0, name1):  # hloc/utils/parsers.py:66-67
    return '_'.join((name0.replace('/', '-'), name1.replace('/', '-')))


def unique_pairs(pair_list):
    """hloc/match_features.py:87-97: skip (a,b) when (a,b) or (b,a) was already matched."""
    matched, out = set(), []
    for pair in pair_list:
        name0, name1 = pair.split(' ')
        if len({(name0, name1), (name1, name0)} & matched):
            continue
        out.append((name0, name1))
        matched |= {(name0, name1), (name1, name0)}
    return out


def cast_for_storage(matches0, scores0):
    """hloc/match_features.py:114,118: matches0 -> int16 (torch .short(): wraps), scores -> fp16."""
    m = np.asarray(matches0).astype(np.int64).astype(np.int16)
    s = np.asarray(scores0).astype(np.float32).astype(np.float16)
    return m, s


def match_pair(model, feats0, feats1):
    """feats*: the per-image groups of the feature file ('descriptors' [128,N], ...).
    Returns (matches0 int16 [N], matching_scores0 fp16 [N]) as the reference stores them."""
    data = {'descriptors0': np.asarray(feats0['descriptors'], dtype=np.float32)[None],
            'descriptors1': np.asarray(feats1['descriptors'], dtype=np.float32)[None]}
    pred = model(data)
    return cast_for_storage(pred['matches0'][0], pred['matching_scores0'][0])


def load_matcher(conf):
    Model = dynamic_load(matchers, conf['model']['name'])   # hloc/match_features.py:77-79
    return Model(conf['model']).eval().to('cuda')


def _out_path(export_dir, features, conf, pairs_name, rank, world):
    base = os.path.join(str(export_dir), f'{features}-{conf["output"]}-{pairs_name}')
    return base + '.h5' if world == 1 else f'{base}.part{rank}of{world}.h5'


def main(conf, pair_list, features, export_dir, pairs_name='pairs', world=1, rank=0, barrier=None, model=None):
    """hloc/match_features.py:48-123: ``pair_list`` holds the pairs file's lines ("name0 name1"),
    ``features`` the feature store's name inside export_dir (:52-54).  Skips pairs already matched in
    either order or already stored (:88-97), writes matches0 int16 / matching_scores0 fp16 per pair
    (:108-116) into <features>-<conf output>-<pairs_name> (:81-82).

    Multi-GPU (:90 is the loop that shards): the de-duplicated pair list is dealt round-robin over
    `world` processes (one per GPU); each writes a part store, and after ``barrier()`` rank 0 merges the parts in
    pair order.  ``model``: a ready matcher (tests inject a stub on CPU); default = the conf's HIP matcher."""
    import json
    from .feature_io import open_store, write_matches
    from .sharding import shard_indices
    if model is None:
        model = load_matcher(conf)
    feats = open_store(os.path.join(str(export_dir), features + '.h5'), 'r')
    out_path = _out_path(export_dir, features, conf, pairs_name, rank, world)
    store = open_store(out_path, 'a' if world == 1 else 'w')
    pairs = unique_pairs(pair_list)
    done = []
    try:
        for idx in shard_indices(len(pairs), rank, world):
            name0, name1 = pairs[idx]
            pair = names_to_pair(name0, name1)
            if pair in store:
                continue
            f0, f1 = feats[name0], feats[name1]
            data = {'descriptors0': np.asarray(f0['descriptors'].__array__(), dtype=np.float32)[None],
                    'descriptors1': np.asarray(f1['descriptors'].__array__(), dtype=np.float32)[None]}
            pred = model(data)
            write_matches(store, pair, pred['matches0'][0], pred['matching_scores0'][0])
            done.append((idx, pair))
        actual = getattr(store, 'path', getattr(store, 'filename', out_path))
    finally:
        store.close()
        feats.close()
    if world == 1:
        return actual
    with open(out_path + '.index.json', 'w') as f:
        json.dump(done, f)
    if barrier is not None:
        barrier()
    if rank != 0:
        return actual
    items = []
    for r in range(world):
        with open(_out_path(export_dir, features, conf, pairs_name, r, world) + '.index.json') as f:
            items += [(int(i), pair, r) for i, pair in json.load(f)]
    items.sort()
    final = open_store(_out_path(export_dir, features, conf, pairs_name, 0, 1), 'a')
    parts = [open_store(_out_path(export_dir, features, conf, pairs_name, r, world), 'r') for r in range(world)]
    try:
        for _, pair, r in items:
            if pair in final:
                continue
            g = final.create_group(pair)
            for k in parts[r][pair].keys():
                g.create_dataset(k, data=np.asarray(parts[r][pair][k].__array__()))   # already int16 / fp16
        actual = getattr(final, 'path', getattr(final, 'filename', None))
    finally:
        final.close()
        for p in parts:
            p.close()
    return actual
