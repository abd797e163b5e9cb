import sqlite3, sys, glob
path = sys.argv[1]
con = sqlite3.connect(path); cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
t = [x for x in tables if "kernel_dispatch" in x and "rocpd" in x][0]
sym = [x for x in tables if "kernel_symbol" in x][0]
scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
nc = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
rows = list(cur.execute(f"select d.start, d.end, s.{nc} from {t} d join {sym} s on d.kernel_id = s.id order by d.start"))
rows = rows[len(rows) // 2:]          # the second repetition
busy = sum(e - s for s, e, _ in rows); wall = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}, wall {wall/1e6:.2f} ms, busy {busy/1e6:.2f} ms ({100*busy/wall:.1f} %)")
from collections import defaultdict
g = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = s1 - e0
    k = (n0.split("(")[0][:40], n1.split("(")[0][:40])
    g[k][0] += max(gap, 0); g[k][1] += 1
for k, (tot, cnt) in sorted(g.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{tot/1e3/ max(cnt,1):8.1f} us avg x {cnt:4d} = {tot/1e6:7.2f} ms   {k[0]} -> {k[1]}")
# memory copies
mc = [x for x in tables if "memory_copy" in x and "rocpd" in x]
if mc:
    cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
    print("memory copy table", mc[0], cols[:12])
    r = list(cur.execute(f"select start, end, size from {mc[0]} order by start"))
    r = [x for x in r if x[0] >= rows[0][0]]
    print(f"copies {len(r)}, total {sum(e-s for s,e,_ in r)/1e6:.2f} ms, bytes {sum(x[2] for x in r)/1e6:.1f} MB")
tot = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = n.split("(")[0][:60]
    tot[k][0] += e - s; tot[k][1] += 1
nimg = max(1, tot[[k for k in tot if "fused_stem" in k][0]][1])
print(f"per image ({nimg} images): kernel time us, launches per image")
for k, (t_, c_) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{t_/1e3/nimg:9.1f} us  {c_/nimg:5.2f}  {k}")
# one image's kernel sequence (the 40th stem onwards)
idx = [i for i, r in enumerate(rows) if "fused_stem" in r[2]]
a, b = idx[40], idx[41]
print("one image, kernel by kernel (start offset us, duration us):")
for s, e, n in rows[a:b]:
    print(f"{(s - rows[a][0]) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n.split('(')[0][:50]}")
r = list(cur.execute(f"select start, end, size from {mc[0]} order by start"))
print("copies overlapping this image:", [(round((x[0] - rows[a][0]) / 1e3, 1), round((x[1] - x[0]) / 1e3, 1), x[2]) for x in r if x[1] > rows[a][0] and x[0] < rows[b][0]])
