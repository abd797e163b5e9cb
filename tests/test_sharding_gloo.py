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
