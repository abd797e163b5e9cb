#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (one directory per counter pass) per kernel.
    python tools/pmc_summary.py gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3 > profiles/rNN_pmc_summary.txt
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import sys


def main(dirs):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in dirs:
        for path in glob.glob(d + "/*counter_collection.csv"):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
    print("# per-kernel averages per dispatch (rocprofv3 --pmc, separate passes per counter group)")
    order = sorted(agg, key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", 0))
    for k in order:
        if k.startswith("__amd") or "at::native" in k:
            continue
        c = agg[k]
        n = {name: max(1, len(v)) for name, v in disp[k].items()}
        line = [f"{k[:100]}"]
        for name in sorted(c):
            line.append(f"    {name:28s} {c[name] / n[name]:16.0f}   (n={n[name]})")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd = 2 * c["FETCH_SIZE"] / n["FETCH_SIZE"] * 1024
            wr = c["WRITE_SIZE"] / n["WRITE_SIZE"] * 1024
            line.append(f"    => HBM-side traffic per launch: read {rd/1e6:.1f} MB (2 x FETCH_SIZE) + write {wr/1e6:.1f} MB = {(rd+wr)/1e6:.1f} MB")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
            line.append(f"    => MFMA pipe busy / SIMD: {c['SQ_VALU_MFMA_BUSY_CYCLES'] / n['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.0f} cycles")
        print("\n".join(line))


if __name__ == "__main__":
    main(sys.argv[1:])
