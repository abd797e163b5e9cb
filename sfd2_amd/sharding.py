"""Multi-GPU layout of the hot path: images / pairs / queries are independent
(extract_localization.py:240, hloc/match_features.py:90, it_loc/localizer.py:87), so the
work list is dealt round-robin to one process per GPU and results are gathered by index.
No collective sits on the data path; torch.distributed (RCCL on the GPU box, gloo in the
CPU tests) is only used for the barrier and the final ordered gather of small results."""


def shard_indices(n_items, rank, world):
    """Indices owned by `rank`: rank, rank + world, ...  (static round-robin)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_items, world))


def merge_ordered(per_rank_results, n_items):
    """per_rank_results[r] = list of (index, payload) produced by rank r -> payloads in index order."""
    out = [None] * n_items
    seen = 0
    for res in per_rank_results:
        for idx, payload in res:
            if out[idx] is not None:
                raise ValueError(f"item {idx} produced twice")
            out[idx] = payload
            seen += 1
    if seen != n_items:
        raise ValueError(f"{n_items - seen} items missing")
    return out


def gather_ordered(local_results, n_items, dist=None):
    """All ranks contribute [(index, payload)]; rank 0 gets the ordered list (others None)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_ordered([local_results], n_items)
    world = dist.get_world_size()
    gathered = [None] * world if dist.get_rank() == 0 else None
    dist.gather_object(local_results, gathered, dst=0)
    if dist.get_rank() != 0:
        return None
    return merge_ordered(gathered, n_items)
