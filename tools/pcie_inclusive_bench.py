"""PCIe-inclusive extraction rate: host images -> key points, the way extract_localization.main feeds the device.
Compares the decoder's uint8 HWC image (5.8 MB at 1600x1200) with the reference's float32 CHW tensor (23 MB), pageable vs
pinned host memory, synchronous calls vs SFD2_FLAG_ASYNC calls issued back to back (the copy stream uploads image i + 1
while image i runs).  Never part of bench.py's `value` (inputs there are resident in HBM)."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2

H, W, K, N = 1200, 1600, 4096, 40
PREC = sys.argv[1] if len(sys.argv) > 1 else "f16c"      # python tools/pcie_inclusive_bench.py [f16c | f16 | f16x3]
m = ResSegNetV2(outdim=128, require_stability=True, precision=PREC).eval()
m.load_state_dict(synth.make_state_dict(0))
m.cuda(0)
ctx = m.context
img = synth.make_image(H, W, 5)
u8 = np.ascontiguousarray((img.transpose(1, 2, 0) * 255).astype(np.uint8))
kp = torch.zeros((K, 2), device="cuda"); sc = torch.zeros((K,), device="cuda"); de = torch.zeros((K, 128), device="cuda")
n = ctypes.c_int()


def run(host, flags, label):
    ptr = host.ctypes.data if isinstance(host, np.ndarray) else host.data_ptr()
    for _ in range(3):
        _lib.check(ctx.lib.sfd2_extract(ctx.h, ptr, 0, H, W, 0.001, K, flags, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()
    t = time.perf_counter()
    for _ in range(N):
        _lib.check(ctx.lib.sfd2_extract(ctx.h, ptr, 0, H, W, 0.001, K, flags, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
    ctx.sync()
    dt = (time.perf_counter() - t) / N
    print(f"{label:52s} {1e3 * dt:7.3f} ms/image  {1 / dt:7.1f} images/s")


A, U = _lib.FLAG_ASYNC, _lib.FLAG_IMG_U8_HWC
run(img, 0, "float32 CHW, pageable, synchronous")
run(img, A, "float32 CHW, pageable, async calls")
pin_f = torch.from_numpy(img).pin_memory()
run(pin_f, A, "float32 CHW, pinned, async calls")
run(u8, U, "uint8 HWC, pageable, synchronous")
run(u8, U | A, "uint8 HWC, pageable, async calls")
pin_u = torch.from_numpy(u8).pin_memory()
run(pin_u, U | A, "uint8 HWC, pinned, async calls")
dev = torch.from_numpy(img).cuda()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N):
    _lib.check(ctx.lib.sfd2_extract(ctx.h, dev.data_ptr(), 1, H, W, 0.001, K, A, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
ctx.sync()
dt = (time.perf_counter() - t) / N
print(f"{'device-resident input (bench.py --extract-only)':52s} {1e3 * dt:7.3f} ms/image  {1 / dt:7.1f} images/s")
