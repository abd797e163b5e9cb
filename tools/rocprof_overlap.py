#!/usr/bin/env python3
"""Concurrency of the kernel trace in a rocprofv3 results database (two streams per GPU): how much of the busy time has
one kernel on the device and how much has two or more, and which kernels are stretched when they share the device.
    python tools/rocprof_overlap.py DIR/NAME_results.db [first_fraction_to_skip=0.3]"""
import sqlite3
import sys
from collections import defaultdict


def main(path, skip=0.3):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if "kernel_dispatch" in t and "rocpd" in t]
    if not disp:
        print("no kernel dispatch table in", tables)
        return
    t = disp[0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({t})")]
    sym = [x for x in tables if "kernel_symbol" in x][0]
    scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
    rows = list(cur.execute(f"select d.start, d.end, s.{name_col} from {t} d join {sym} s on d.kernel_id = s.id order by d.start"))
    if not rows:
        print("empty trace")
        return
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print("# ten slices of the trace: busy depth shares (0 / 1 / 2+ kernels on the device), launches")
    for i in range(10):
        a, b = t0 + (t1 - t0) * i / 10, t0 + (t1 - t0) * (i + 1) / 10
        evs = []
        for s_, e_, _ in rows:
            if e_ <= a or s_ >= b:
                continue
            evs.append((max(s_, a), 1)); evs.append((min(e_, b), -1))
        evs.sort()
        d, last, h = 0, a, [0, 0, 0]
        for ts, dd in evs:
            h[min(d, 2)] += ts - last; last = ts; d += dd
        h[0] += b - last
        print(f"  slice {i}: {h[0] / (b - a):5.1%} {h[1] / (b - a):5.1%} {h[2] / (b - a):5.1%}   {len(evs) // 2} launches")
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= cut]
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last = 0, ev[0][0]
    hist = defaultdict(int)
    for ts, d in ev:
        hist[depth] += ts - last
        last = ts
        depth += d
    span = ev[-1][0] - ev[0][0]
    print(f"# {path}: {len(rows)} launches after the first {skip:.0%} of the trace, span {span / 1e6:.2f} ms")
    for k in sorted(hist):
        print(f"  {k} kernel(s) on the device: {hist[k] / span:6.1%} of the span")
    dur = defaultdict(list)
    for s, e, n in rows:
        dur[n.split("(")[0][:60]].append((e - s) / 1e3)
    print("# per kernel: launches, mean us in this (concurrent) trace")
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
        print(f"  {len(v):6d} {sum(v) / len(v):9.1f}  {n}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
