"""Per-rank CPU placement (sfd2_amd/sharding.py pin_to_gpu_socket; VERDICT r5 #5a): the cpulist parser, the share a rank gets when several GPUs hang off one
socket, and the pin itself against a fake sysfs tree -- no GPU: the PCI addresses are handed in."""
import os

import pytest

from sfd2_amd import sharding as sh


def test_parse_cpulist():
    assert sh.parse_cpulist("0-3,8-11\n") == [0, 1, 2, 3, 8, 9, 10, 11]
    assert sh.parse_cpulist("5") == [5] and sh.parse_cpulist("") == [] and sh.parse_cpulist("2,0-1") == [0, 1, 2]


def test_shares_on_a_two_socket_eight_gpu_host():
    """2 x 64 cores, siblings numbered + 128 (the GPU box's EPYC 9575F numbering), four GPUs per socket: every rank gets 16 cores AND their 16 siblings, disjoint
    from every other rank, inside its own socket."""
    s0 = list(range(0, 64)) + list(range(128, 192))
    s1 = list(range(64, 128)) + list(range(192, 256))
    lists = [s0] * 4 + [s1] * 4
    shares = [sh.share_of_cpus(lists, r) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(c for s in shares for c in s) == list(range(256))
    for r, s in enumerate(shares):
        assert set(s) <= set(lists[r])
        cores = [c for c in s if c < 128]
        assert sorted(c + 128 for c in cores) == [c for c in s if c >= 128]          # a core and its sibling stay together
    # one rank per socket keeps the whole socket; the current mask (a container's cpuset) is intersected in
    assert sh.share_of_cpus([s0, s1], 1) == s1
    assert sh.share_of_cpus([s0, s1], 0, allowed=range(0, 8)) == list(range(0, 8))
    assert sh.share_of_cpus([[], s1], 0) == []                                       # unknown: do not pin


def _fake_sysfs(root, devices):
    for bus, (cpulist, node) in devices.items():
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        if cpulist is not None:
            (d / "local_cpulist").write_text(cpulist + "\n")
        (d / "numa_node").write_text(f"{node}\n")
    for node, cpulist in {0: "0-1", 1: "2-3"}.items():
        n = root / "devices" / "system" / "node" / f"node{node}"
        n.mkdir(parents=True)
        (n / "cpulist").write_text(cpulist + "\n")


def test_local_cpus_of_pci_reads_cpulist_then_numa_node(tmp_path):
    _fake_sysfs(tmp_path, {"0000:05:00.0": ("0-1", 0), "0000:85:00.0": (None, 1), "0000:c5:00.0": ("", -1)})
    assert sh.local_cpus_of_pci("0000:05:00.0", str(tmp_path)) == [0, 1]
    assert sh.local_cpus_of_pci("0000:85:00.0", str(tmp_path)) == [2, 3]             # no local_cpulist: the NUMA node's
    assert sh.local_cpus_of_pci("0000:C5:00.0", str(tmp_path)) == []                 # neither (upper-case address as HIP prints it)
    assert sh.local_cpus_of_pci("0000:ff:00.0", str(tmp_path)) == []


@pytest.mark.skipif(not hasattr(os, "sched_setaffinity"), reason="no sched_setaffinity")
def test_pin_to_gpu_socket_sets_and_reports_the_mask(tmp_path):
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        pytest.skip("one CPU")
    a, b = have[0], have[1]
    _fake_sysfs(tmp_path, {"0000:05:00.0": (f"{a}", 0), "0000:85:00.0": (f"{b}", 1), "0000:c5:00.0": ("", -1)})
    bus = ["0000:05:00.0", "0000:85:00.0", "0000:c5:00.0"]
    try:
        info = sh.pin_to_gpu_socket(1, local_world=3, enable=True, sysfs=str(tmp_path), bus_ids=bus)
        assert info["pinned"] and info["cpus"] == [b] and info["bus_id"] == bus[1]
        assert sorted(os.sched_getaffinity(0)) == [b]
        os.sched_setaffinity(0, have)
        info = sh.pin_to_gpu_socket(2, local_world=3, enable=True, sysfs=str(tmp_path), bus_ids=bus)      # unknown locality: floats as before
        assert not info["pinned"] and sorted(os.sched_getaffinity(0)) == have
        info = sh.pin_to_gpu_socket(0, enable=False, sysfs=str(tmp_path), bus_ids=bus)
        assert not info["pinned"] and info["why"] == "disabled"
        os.environ["SFD2_CPU_AFFINITY"] = "0"
        assert sh.pin_to_gpu_socket(0, sysfs=str(tmp_path), bus_ids=bus)["why"] == "disabled"
    finally:
        os.environ.pop("SFD2_CPU_AFFINITY", None)
        os.sched_setaffinity(0, have)


def test_pin_without_a_device_reports_and_does_not_raise():
    """On a host without a GPU (this test's usual home) the library has no PCI address to give: pinned False, the reason in 'why'."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    info = sh.pin_to_gpu_socket(0, enable=True)
    assert info["pinned"] is False and "no device" in info["why"]
