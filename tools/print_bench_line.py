"""One-line summary of a bench.py JSON line: python tools/print_bench_line.py <file.json> [tag]   (or: ... - tag < file)."""
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "-"
text = open(src).read() if src != "-" and os.path.exists(src) else (sys.stdin.read() if not sys.stdin.isatty() else "")
d = json.loads(text)
print(sys.argv[2] if len(sys.argv) > 2 else src, d["value"], d["ms_per_step"], d.get("device_ms_per_step"), d["single_stream"]["value"])
