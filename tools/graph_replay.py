"""hipGraph capture of one extract + match step (BASELINE configs[4] asks for per-GPU hipGraph capture).
The library's calls are plain stream work when device-resident inputs/outputs and SFD2_FLAG_ASYNC are used, so the
caller can capture them on the context's stream (sfd2_get_stream) and replay the graph.  Prints eager vs replayed
step time and checks the replayed outputs against the eager ones."""
import ctypes
import sys
import time

import torch

sys.path.insert(0, ".")
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2

hip = ctypes.CDLL("libamdhip64.so")
H, W, K, KDB, N = 1200, 1600, 4096, 50, 60
dev = torch.device("cuda", 0)
m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval()
m.load_state_dict(synth.make_state_dict(0))
m.cuda(0)
ctx = m.context
lib = ctx.lib
img = torch.from_numpy(synth.make_image(H, W, 100)).to(dev)
g = torch.Generator(device="cpu").manual_seed(1234)
db = []
for _ in range(KDB):
    d = torch.randn(K, 128, generator=g)
    db.append((d / d.norm(dim=1, keepdim=True)).to(torch.float16).to(dev).contiguous())
dbs = (_lib.DescSet * KDB)(*[_lib.DescSet(d.data_ptr(), K, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in db])
mconf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)
kp = torch.zeros((K, 2), device=dev); sc = torch.zeros((K,), device=dev); de = torch.zeros((K, 128), device=dev)
mt = torch.zeros((KDB, K), dtype=torch.int64, device=dev); ms = torch.zeros((KDB, K), device=dev)
q = _lib.DescSet(de.data_ptr(), K, _lib.DT_F32, _lib.LAYOUT_ND, 1)
n = ctypes.c_int()


def step():
    _lib.check(lib.sfd2_extract(ctx.h, img.data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(),
                                de.data_ptr(), 1, K, ctypes.byref(n)))
    _lib.check(lib.sfd2_match_batch(ctx.h, ctypes.byref(q), dbs, KDB, 128, ctypes.byref(mconf), mt.data_ptr(), ms.data_ptr(), 1,
                                    _lib.FLAG_ASYNC))


for _ in range(3):
    step()
ctx.sync()
t = time.perf_counter()
for _ in range(N):
    step()
ctx.sync()
eager = (time.perf_counter() - t) / N
ref = (kp.clone(), sc.clone(), mt.clone())

lib.sfd2_get_stream.restype = ctypes.c_void_p
lib.sfd2_get_stream.argtypes = [ctypes.c_void_p]
stream = ctypes.c_void_p(lib.sfd2_get_stream(ctx.h))
graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
rc = hip.hipStreamBeginCapture(stream, 2)          # hipStreamCaptureModeRelaxed
assert rc == 0, rc
step()
rc = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
assert rc == 0 and graph.value, rc
rc = hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0)
assert rc == 0, rc
kp.zero_(); sc.zero_(); mt.zero_()
torch.cuda.synchronize()
for _ in range(3):
    assert hip.hipGraphLaunch(gexec, stream) == 0
ctx.sync()
t = time.perf_counter()
for _ in range(N):
    hip.hipGraphLaunch(gexec, stream)
ctx.sync()
replay = (time.perf_counter() - t) / N
ok = bool((kp == ref[0]).all() and (sc == ref[1]).all() and (mt == ref[2]).all())
print(f"eager  {1e3 * eager:.3f} ms/step  {1 / eager:.1f} images/s")
print(f"graph  {1e3 * replay:.3f} ms/step  {1 / replay:.1f} images/s   outputs identical: {ok}")
