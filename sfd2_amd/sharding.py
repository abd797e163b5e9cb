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


# ---------------------------------------------------------------------------------------------- CPU placement of a rank
# One process per GPU, each with a decoder pool (16 threads by default) and writer threads: on an 8-GPU node that is 128 + 24 threads on a two-socket host.
# Left to the scheduler they float over both sockets and half of the pinned-buffer traffic crosses the socket link.  A rank therefore pins itself (and
# the threads it starts afterwards: they inherit the mask) to the CPUs of the socket its GPU hangs off -- sysfs `local_cpulist` of the GPU's PCI device --
# and, when several ranks share a socket, to its own contiguous share of those CPUs.  The reference's shape is 4 DataLoader worker processes per
# extracting process, unplaced (extract_localization.py:230-233).
def parse_cpulist(text):
    """'0-63,128-191' -> [0, ..., 63, 128, ..., 191] (the kernel's cpulist format; empty -> [])."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def local_cpus_of_pci(bus_id, sysfs="/sys"):
    """CPUs local to a PCI device ('0000:c1:00.0'): its local_cpulist, else its NUMA node's cpulist, else [] (unknown)."""
    import os
    dev = os.path.join(sysfs, "bus", "pci", "devices", bus_id.lower())
    try:
        with open(os.path.join(dev, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        if cpus:
            return cpus
    except OSError:
        pass
    try:
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
        if node >= 0:
            with open(os.path.join(sysfs, "devices", "system", "node", f"node{node}", "cpulist")) as f:
                return parse_cpulist(f.read())
    except (OSError, ValueError):
        pass
    return []


def share_of_cpus(cpu_lists, rank, allowed=None):
    """cpu_lists[r] = CPUs local to rank r's GPU.  Ranks with the SAME list split it into contiguous shares in rank order (hyper-thread siblings, which
    the kernel numbers cpu and cpu + n_cores, end up in the same share when the list is 'a-b,c-d' with equal halves: a share takes the same slice of
    both halves).  `allowed` (the process's current mask, e.g. a container's cpuset) is intersected in.  [] = unknown: do not pin."""
    mine = list(cpu_lists[rank])
    if allowed is not None:
        mine = [c for c in mine if c in set(allowed)]
    if not mine:
        return []
    peers = [r for r, l in enumerate(cpu_lists) if list(l) == list(cpu_lists[rank])]
    k, n = peers.index(rank), len(peers)
    if n == 1:
        return mine
    # runs of consecutive CPU numbers (physical cores first, their siblings as a second run on the usual numbering)
    runs, cur = [], [mine[0]]
    for c in mine[1:]:
        if c == cur[-1] + 1:
            cur.append(c)
        else:
            runs.append(cur)
            cur = [c]
    runs.append(cur)
    out = []
    for run in runs:
        lo, hi = len(run) * k // n, len(run) * (k + 1) // n
        out.extend(run[lo:hi])
    return out or mine


def pin_to_gpu_socket(local_rank, local_world=None, enable=None, sysfs="/sys", bus_ids=None):
    """Pins the calling process (os.sched_setaffinity; threads started later inherit) to the CPUs of the socket GPU `local_rank` hangs off, and to this
    rank's share of them when `local_world` ranks of the node share sockets.  Returns {"cpus": [...], "bus_id", "pinned": bool, "why"} -- never raises for
    a host it cannot read (no sysfs entry, no sched_setaffinity): the run then floats as before.  enable=None reads SFD2_CPU_AFFINITY (default on; "0" = off).
    bus_ids: the PCI addresses by device index (default: asked from the library, sfd2_device_pci_bus_id)."""
    import os
    if enable is None:
        enable = os.environ.get("SFD2_CPU_AFFINITY", "1") not in ("0", "off", "no")
    info = {"cpus": [], "bus_id": None, "pinned": False, "why": ""}
    if not enable:
        info["why"] = "disabled"
        return info
    if not hasattr(os, "sched_setaffinity"):
        info["why"] = "no sched_setaffinity on this platform"
        return info
    if bus_ids is None:
        import ctypes
        from . import _lib
        lib = _lib.load()
        n = ctypes.c_int(0)
        buf = ctypes.create_string_buffer(32)
        if lib.sfd2_device_pci_bus_id(int(local_rank), buf, 32, ctypes.byref(n)) != 0:
            info["why"] = "no device: " + lib.sfd2_last_error().decode()
            return info
        world = n.value if local_world is None else min(int(local_world), n.value)
        bus_ids = []
        for d in range(max(world, local_rank + 1)):
            lib.sfd2_device_pci_bus_id(d, buf, 32, None)
            bus_ids.append(buf.value.decode())
    elif local_world is not None:
        bus_ids = list(bus_ids)[:max(int(local_world), local_rank + 1)]
    info["bus_id"] = bus_ids[local_rank]
    lists = [local_cpus_of_pci(b, sysfs) for b in bus_ids]
    cpus = share_of_cpus(lists, local_rank, allowed=sorted(os.sched_getaffinity(0)))
    if not cpus:
        info["why"] = f"no local_cpulist / numa_node for {bus_ids[local_rank]} inside the current mask"
        return info
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        info["why"] = f"sched_setaffinity: {e}"
        return info
    info.update(cpus=cpus, pinned=True, why=f"{len(cpus)} CPUs local to {bus_ids[local_rank]}")
    return info
