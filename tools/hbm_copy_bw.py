"""Achievable HBM bandwidth on this GPU for the access mixes the HBM-bound layers have
(pure read, 1 read + 1 write, 2 reads + 1 write), measured with torch element-wise ops on
61 MB / 123 MB / 1 GB fp16 tensors.  Gives the practical ceiling the roofline fractions of
the 1x1 / grouped convolutions should be read against (MI355X_MICROARCH.md quotes 8 TB/s peak)."""
import torch

def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

for mb in (61, 123, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device="cuda", dtype=torch.float16)
    b = torch.randn(n, device="cuda", dtype=torch.float16)
    c = torch.empty_like(a)
    by = n * 2
    print(f"{mb:5d} MB  read(sum)   {by / t(lambda: a.sum()) / 1e9:8.0f} GB/s")
    print(f"{mb:5d} MB  copy r+w    {2 * by / t(lambda: c.copy_(a)) / 1e9:8.0f} GB/s")
    print(f"{mb:5d} MB  add 2r+w    {3 * by / t(lambda: torch.add(a, b, out=c)) / 1e9:8.0f} GB/s")
    print(f"{mb:5d} MB  fill w      {by / t(lambda: c.zero_()) / 1e9:8.0f} GB/s")
