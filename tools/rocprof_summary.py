#!/usr/bin/env python3
"""Turns a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd,
ROCm 7.2 writes NAME_results.db) into the per-kernel stats table committed under profiles/.
    python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_rocprof_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats  ({path})  durations in microseconds")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"{calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {short}")


if __name__ == "__main__":
    main(sys.argv[1])
