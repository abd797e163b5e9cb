#!/usr/bin/env python3
"""How many CPUs does this container REALLY get?  `nproc` / sched_getaffinity count the CPUs a process may run on; a cgroup CPU quota (cpu.max) caps the CPU TIME
all processes of the container get together -- on such a box 128 busy threads make no more progress than the quota's worth, and every "all cores" figure
(bench.py's cpu_baseline.all_cores, tools/host_soak.py's aggregate) measures the quota, not the host.  Prints the cgroup's cpu.max / cpu.stat and the measured
aggregate rate of N busy processes (pure-Python spin, no memory traffic) for N = 1 .. logical CPUs."""
import json
import multiprocessing as mp
import os
import time


def spin(seconds, q):
    n = 0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(10000):
            n += 1
    q.put(n)


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def main():
    out = {"logical_cpus": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
           "cgroup_v2_cpu_max": read("/sys/fs/cgroup/cpu.max"), "cgroup_v1_cfs_quota_us": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"),
           "cgroup_v1_cfs_period_us": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"), "cpu_stat_before": read("/sys/fs/cgroup/cpu.stat")}
    rates = {}
    base = None
    n = 1
    counts = []
    while n <= (os.cpu_count() or 1):
        counts.append(n)
        n *= 2
    for n in counts:
        q = mp.Queue()
        ps = [mp.Process(target=spin, args=(2.0, q)) for _ in range(n)]
        t0 = time.perf_counter()
        for p in ps:
            p.start()
        tot = sum(q.get() for _ in ps)
        for p in ps:
            p.join()
        dt = time.perf_counter() - t0
        rate = tot / dt
        base = base or rate
        rates[n] = round(rate / base, 2)
    out["aggregate_speedup_by_processes"] = rates
    out["cpu_stat_after"] = read("/sys/fs/cgroup/cpu.stat")
    eff = max(rates.values())
    out["effective_parallel_cpus"] = eff
    print(json.dumps(out))


if __name__ == "__main__":
    main()
