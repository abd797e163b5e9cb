"""The N>1 path on CPU: two processes (gloo) shard a work list the way bench.py / the drivers
do on N GPUs (round-robin by index, no data-path collective) and rank 0 merges by index."""
import os
import subprocess
import sys
import textwrap

import pytest

from sfd2_amd.sharding import merge_ordered, shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partition():
    for n in (0, 1, 7, 50, 1015):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)
    with pytest.raises(ValueError):
        merge_ordered([[(0, "a")], [(0, "b")]], 1)
    with pytest.raises(ValueError):
        merge_ordered([[(0, "a")]], 2)


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from sfd2_amd.sharding import shard_indices, gather_ordered
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["PORT"],
                            rank=int(os.environ["RANK"]), world_size=2)
    rank = dist.get_rank()
    items = ["img%%03d" %% i for i in range(11)]
    mine = [(i, (items[i], rank)) for i in shard_indices(len(items), rank, 2)]   # stand-in for extract()
    dist.barrier()
    t = torch.tensor([1.0 + rank])           # per-rank elapsed time -> MAX over ranks (bench.py contract)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 2.0
    merged = gather_ordered(mine, len(items), dist)
    if rank == 0:
        assert [m[0] for m in merged] == items
        assert [m[1] for m in merged] == [i %% 2 for i in range(11)]
        print("MERGED_OK")
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_two_process_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "MERGED_OK" in outs[0]


DRIVER_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from sfd2_amd import extract_localization as el, match_features as mf
    from sfd2_amd.feature_io import open_store
    rank, world = int(os.environ["RANK"]), 2
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%%s" %% os.environ["PORT"], rank=rank, world_size=world)
    out = os.environ["OUT"]

    def stub_extractor(model, img, topK, mask, conf_th, scales):      # stands in for the HIP extractor (no GPU here)
        a = np.asarray(img, dtype=np.float64).reshape(-1)
        n = 5 + int(a[0] * 10) %% 4
        rs = np.random.RandomState(int(a[:8].sum() * 1e6) %% (2 ** 31))
        return {"keypoints": rs.rand(n, 2) * 30, "descriptors": rs.rand(n, 128), "scores": np.sort(rs.rand(n))[::-1].copy()}

    class StubMatcher:                                               # hloc matcher interface (dict in, dict out)
        def __call__(self, data):
            d0, d1 = data["descriptors0"][0], data["descriptors1"][0]
            sim = d0.T @ d1
            return {"matches0": sim.argmax(1)[None], "matching_scores0": sim.max(1)[None]}

    images = [{"name": "db/img%%02d.jpg" %% i, "image": np.full((3, 16, 24), (i + 1) / 16.0, np.float32),
               "original_size": (48, 32)} for i in range(7)]
    conf = el.confs["ressegnetv2-20220810-wapv2-sd2mfsf-uspg-0001-n4096-r1600"]
    path = el.main(conf, images, out, world=world, rank=rank, barrier=dist.barrier, model_and_extractor=(None, stub_extractor))
    dist.barrier()
    pairs = ["db/img00.jpg db/img01.jpg", "db/img01.jpg db/img00.jpg", "db/img02.jpg db/img05.jpg", "db/img03.jpg db/img04.jpg",
             "db/img06.jpg db/img00.jpg", "db/img05.jpg db/img06.jpg"]
    mpath = mf.main(mf.confs["NNM"], pairs, conf["output"], out, world=world, rank=rank, barrier=dist.barrier, model=StubMatcher())
    dist.barrier()
    if rank == 0:
        # the merged stores equal what ONE process writes
        solo = os.path.join(out, "solo")
        p1 = el.main(conf, images, solo, model_and_extractor=(None, stub_extractor))
        m1 = mf.main(mf.confs["NNM"], pairs, conf["output"], solo, model=StubMatcher())
        for got, want in ((path, p1), (mpath, m1)):
            a, b = open_store(got, "r"), open_store(want, "r")
            assert list(a.keys()) == list(b.keys()) and len(list(a.keys())) > 0, (list(a.keys()), list(b.keys()))
            for k in a.keys():
                for ds in b[k].keys():
                    x, y = np.asarray(a[k][ds].__array__()), np.asarray(b[k][ds].__array__())
                    assert x.dtype == y.dtype and np.array_equal(x, y), (k, ds)
        assert len(list(open_store(mpath, "r").keys())) == 5          # (img01, img00) is a duplicate of (img00, img01)
        print("DRIVERS_OK")
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_sharded_drivers_two_process_gloo(tmp_path):
    """extract_localization.main and match_features.main with world = 2 (stub extractor / matcher: no GPU here):
    each rank writes its part, rank 0 merges in item order, and the result equals a single-process run."""
    script = tmp_path / "d.py"
    script.write_text(DRIVER_WORKER)
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), PORT=str(port), OUT=str(tmp_path / "out"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "DRIVERS_OK" in outs[0]
