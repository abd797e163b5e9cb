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


def group_pairs(pairs):
    """[(name0, name1)] -> [(name0, [(pair index, name1), ...])] in order of first appearance: the query-owns-its-matches
    unit (every pair of a group shares descriptors0, hloc/match_features.py:99-103)."""
    groups, order = {}, []
    for idx, (name0, name1) in enumerate(pairs):
        if name0 not in groups:
            groups[name0] = []
            order.append(name0)
        groups[name0].append((idx, name1))
    return [(name0, groups[name0]) for name0 in order]


def _match_grouped(model, feats, store, pairs, rank, world, done, device=0, max_batch=64, readers=4, cache_bytes=None,
                   lookahead=8):
    """The per-pair loop of main() with the reference's read + .float() + .cuda() per pair (hloc/match_features.py:99-105)
    replaced by device-resident sets: pairs are grouped by their first image, every descriptor set is converted once
    (sfd2_desc_pack) and kept in HBM as fp16 [n][128] (LRU, sfd2_amd/pipeline.py ResidentSets), one sfd2_match_batch per
    group (at most `max_batch` pairs per launch), results cast and appended by a writer thread while the next group
    runs.  Sharding (world > 1) is by group: the process that owns a query runs all its pairs.  Writes the same
    groups as the per-pair loop (same conversion kernel, same matcher kernels, same casts)."""
    import ctypes
    import torch
    from . import _lib
    from .feature_io import write_matches
    from .pipeline import ResidentSets, WriterPool
    from .sharding import shard_indices
    ctx = _lib.default_context(device)
    conf = model.match_conf()
    groups = group_pairs(pairs)
    mine = [groups[g] for g in shard_indices(len(groups), rank, world)]
    # split into launches of <= max_batch pairs, skipping pairs the store already holds (hloc/match_features.py:93-94)
    units = []
    for name0, members in mine:
        todo = [(idx, name1) for idx, name1 in members if names_to_pair(name0, name1) not in store]
        for i in range(0, len(todo), max_batch):
            units.append((name0, todo[i:i + max_batch]))
    if not units:
        return
    sets = ResidentSets(ctx, feats, budget=cache_bytes, readers=readers)
    dev = torch.device("cuda", ctx.device)
    stream = torch.cuda.ExternalStream(ctx.stream, device=dev)

    class Ring:
        def __init__(self):
            self.m = self.s = None
            self.event = torch.cuda.Event()

        def reserve(self, k, n0):
            # int16 / fp16: the stored types, written by the matcher itself (SFD2_FLAG_MATCH_OUT16)
            if self.m is None or self.m.numel() < k * n0:
                self.m = torch.empty(max(k * n0, 1), dtype=torch.int16, pin_memory=True)
                self.s = torch.empty(max(k * n0, 1), dtype=torch.float16, pin_memory=True)
            return self.m.numpy()[:k * n0].reshape(k, n0), self.s.numpy()[:k * n0].reshape(k, n0)

    import queue
    free = queue.Queue()
    for _ in range(3):
        free.put(Ring())
    lock = __import__("threading").Lock()

    def write(job):
        ring, m, s, name0, members = job
        try:
            ring.event.synchronize()
            m16, s16 = m, s                            # already int16 / fp16 as stored (:114,118): the device did the casts
            pairs = [names_to_pair(name0, name1) for _, name1 in members]
            if hasattr(store, "write_rows"):
                store.write_rows(pairs, {"matches0": m16[:len(pairs)], "matching_scores0": s16[:len(pairs)]})      # the query's pair groups as ONE append
            else:
                for i, pair in enumerate(pairs):
                    if hasattr(store, "write_group"):
                        store.write_group(pair, {"matches0": m16[i], "matching_scores0": s16[i]})
                    else:
                        write_matches(store, pair, m16[i], s16[i])
        finally:
            free.put(ring)                             # (the store has copied the rows -- or failed: either way the producer must not wait for this ring for ever)
        with lock:
            done.extend((idx, pair) for (idx, _), pair in zip(members, pairs))

    def drop(job):                                     # queued behind a failed write: not written, its ring still goes back
        job[0].event.synchronize()
        free.put(job[0])

    wp = WriterPool(write, workers=1, maxsize=2, name="sfd2-match-writer", on_drop=drop)
    db_arrays = {}
    try:
        for u, (name0, members) in enumerate(units):
            # reads of the units ahead: the whole window once, then only the unit that enters it (a set evicted in between is read again by get())
            for name0_next, members_next in (units[:1 + lookahead] if u == 0 else units[u + lookahead:u + lookahead + 1]):
                sets.prefetch([name0_next] + [n1 for _, n1 in members_next])
            q_ptr, n0 = sets.get(name0, u)
            k = len(members)
            db = db_arrays.get(k)             # (the library reads the array during the call: reusable afterwards)
            if db is None:
                db = db_arrays[k] = (_lib.DescSet * k)(*[_lib.DescSet(None, 0, _lib.DT_F16, _lib.LAYOUT_ND, 1, None, 0, 0) for _ in range(k)])
            get = sets.get
            for i, (_, name1) in enumerate(members):
                e = db[i]
                e.data, e.n = get(name1, u)
            q = _lib.DescSet(q_ptr, n0, _lib.DT_F16, _lib.LAYOUT_ND, 1, None, 0, 0)
            ring = free.get()
            m, s = ring.reserve(k, n0)
            if n0 == 0:
                pass                                 # nothing to match: empty rows, as the per-pair call returns
            else:
                _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(q), db, k, 128, ctypes.byref(conf), m.ctypes.data,
                                                    s.ctypes.data, 0, _lib.FLAG_ASYNC | _lib.FLAG_MATCH_OUT16))
            ring.event.record(stream)
            wp.put((ring, m, s, name0, members))
            sets.completed_seq = u - 3               # three rings: a unit's buffers are reused only after its event was awaited
    finally:
        wp.close()
        ctx.sync()
        sets.close()
    done.sort()
    return {"loads": sets.loads, "hits": sets.hits, "evictions": sets.evictions}


def _out_path(export_dir, features, conf, pairs_name, rank, world):
    base = os.path.join(str(export_dir), f'{features}-{conf["output"]}-{pairs_name}')
    return base + '.h5' if world == 1 else f'{base}.part{rank}of{world}.h5'


def main(conf, pair_list, features, export_dir, pairs_name='pairs', world=1, rank=0, barrier=None, model=None,
         grouped=None, device=0, cache_bytes=None, readers=4, affinity=None):
    """hloc/match_features.py:48-123: ``pair_list`` holds the pairs file's lines ("name0 name1"),
    ``features`` the feature store's name inside export_dir (:52-54).  Skips pairs already matched in
    either order or already stored (:88-97), writes matches0 int16 / matching_scores0 fp16 per pair
    (:108-116) into <features>-<conf output>-<pairs_name> (:81-82).

    Multi-GPU (:90 is the loop that shards): the de-duplicated pair list is dealt round-robin over
    `world` processes (one per GPU); each writes a part store, and after ``barrier()`` rank 0 merges the parts in
    pair order.  ``model``: a ready matcher (tests inject a stub on CPU); default = the conf's HIP matcher.

    grouped (default: on for the HIP nearest-neighbour matcher, off for an injected model): _match_grouped above --
    device-resident descriptor sets and one batched launch per query instead of the reference's read / cast / upload /
    launch per pair; shards by query group instead of by pair.  grouped=False is the reference's loop.  Same store."""
    import json
    from .feature_io import open_store, write_matches
    from .sharding import pin_to_gpu_socket, shard_indices
    if affinity or (affinity is None and world > 1):     # before the reader / writer threads exist (extract_localization.main has the same switch)
        pin_to_gpu_socket(device, local_world=world if world > 1 else None, enable=True if affinity else None)
    if grouped is None:
        grouped = model is None and conf['model']['name'] == 'nearest_neighbor'
    if model is None:
        model = load_matcher(conf)
    feats = open_store(os.path.join(str(export_dir), features + '.h5'), 'r')
    out_path = _out_path(export_dir, features, conf, pairs_name, rank, world)
    store = open_store(out_path, 'a' if world == 1 else 'w')
    pairs = unique_pairs(pair_list)
    done = []
    try:
        if grouped:
            _match_grouped(model, feats, store, pairs, rank, world, done, device=device, cache_bytes=cache_bytes, readers=readers)
        for idx in (() if grouped else shard_indices(len(pairs), rank, world)):
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


def cli(argv=None):
    """The reference script's command line (hloc/match_features.py:129-142: --export_dir, --features, --pairs, --conf, --exhaustive).  --exhaustive writes
    every unordered pair of the feature store's images to --pairs first (:61-74; image names in sorted order here, where the reference takes them out of a
    set).  Under torchrun every rank matches its share of the query groups on its own GPU and rank 0 merges."""
    import argparse
    from pathlib import Path
    from .extract_localization import _dist_env
    from .feature_io import open_store
    ap = argparse.ArgumentParser(description="SFD2 descriptor matching on MI355X (drop-in for the reference's hloc/match_features.py)")
    ap.add_argument('--export_dir', type=Path, required=True)
    ap.add_argument('--features', type=str, default='feats-ressegnetv2-20220810-wapv2-sd2mfsf-uspg-0001-n4096-r1600')
    ap.add_argument('--pairs', type=Path, required=True)
    ap.add_argument('--conf', type=str, default='NNM', choices=list(confs.keys()))
    ap.add_argument('--exhaustive', action='store_true')
    ap.add_argument('--device', type=int, default=None)
    args = ap.parse_args(argv)
    world, rank, local, barrier = _dist_env()
    device = args.device
    if device is None:
        import torch
        device = local % max(1, torch.cuda.device_count())
    if args.exhaustive:
        if rank == 0:
            assert not args.pairs.exists(), args.pairs
            feats = open_store(os.path.join(str(args.export_dir), args.features + '.h5'), 'r')
            images = sorted(feats.keys())
            feats.close()
            with open(str(args.pairs), 'w') as f:
                f.write('\n'.join(' '.join((images[i], images[j])) for i in range(len(images)) for j in range(i)))
        if barrier is not None:
            barrier()
    assert args.pairs.exists(), args.pairs
    with open(args.pairs, 'r') as f:
        pair_list = f.read().rstrip('\n').split('\n')
    path = main(confs[args.conf], pair_list, args.features, args.export_dir, pairs_name=args.pairs.stem, world=world, rank=rank, barrier=barrier, device=device)
    if rank == 0:
        print(path)
    return path


if __name__ == '__main__':
    cli()
