import sys, ctypes, time, collections
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
from sfd2_amd import _lib
import torch, numpy as np
from sfd2_amd import synth
from sfd2_amd.model import ResSegNetV2
m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
ctx = m.context; lib = ctx.lib
H, W, K = 1200, 1600, 4096
NB = 8
u8_host = [torch.from_numpy((synth.make_image(H, W, 100 + i).transpose(1, 2, 0) * 255).astype(np.uint8).copy()).pin_memory() for i in range(NB)]
u8_dev = [t.cuda() for t in u8_host]
outs_d = [(torch.empty((K, 2), device="cuda"), torch.empty((K,), device="cuda"), torch.empty((K, 128), device="cuda")) for _ in range(NB)]
outs_h = [(torch.empty((K, 2)).pin_memory(), torch.empty((K,)).pin_memory(), torch.empty((K, 128)).pin_memory()) for _ in range(NB)]
rec = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(NB)]
stream = torch.cuda.ExternalStream(ctx.stream)
n = ctypes.c_int()
def run(reps, in_host, out_host, depth, record):
    q = collections.deque()
    for i in range(reps):
        b = i % NB
        src, on = (u8_host[b].data_ptr(), 0) if in_host else (u8_dev[b].data_ptr(), 1)
        kp, sc, de = outs_h[b] if out_host else outs_d[b]
        _lib.check(lib.sfd2_extract(ctx.h, src, on, H, W, 0.001, K, _lib.FLAG_ASYNC | _lib.FLAG_IMG_U8_HWC, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 0 if out_host else 1, K, ctypes.byref(n)))
        if record:
            _lib.check(lib.sfd2_extract_record_async(ctx.h, rec[b].data_ptr(), 0))
        ev = torch.cuda.Event(); ev.record(stream); q.append(ev)
        if len(q) >= depth:
            q.popleft().synchronize()
    ctx.sync()
for in_host, out_host, depth, record in ((False, False, 64, False), (False, False, 3, False), (True, True, 64, False), (True, True, 3, False), (True, True, 3, True), (True, False, 3, False), (False, True, 3, False), (True, True, 1, True)):
    run(24, in_host, out_host, depth, record)
    t0 = time.perf_counter(); run(96, in_host, out_host, depth, record); ms = (time.perf_counter() - t0) / 96 * 1e3
    ctx.set_profiling(40); run(32, in_host, out_host, depth, record); rows = ctx.layer_timings(); ctx.set_profiling(0)
    d = {r["name"]: round(r["ms_total"] / max(1, r["launches"]) * 1e3, 1) for r in rows}
    print(f"in {'host' if in_host else 'dev '} out {'host' if out_host else 'dev '} depth {depth:2d} record {int(record)}: {ms:.3f} ms/extract; stem {d.get('conv1a+conv1b')} conv2b {d.get('conv2b')} conv4.0.conv1 {d.get('conv4.0.conv1')} rb23 {d.get('conv4.0.conv3')} conv3b {d.get('conv3b')}", flush=True)
