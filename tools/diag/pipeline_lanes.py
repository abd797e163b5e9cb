"""Where does the pipelined extract loop spend its time?  Pre-decoded uint8 images (no JPEG decode), with and without the writers' work."""
import sys, time, os, tempfile
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", ".."))
import numpy as np
from sfd2_amd import synth, extract_localization as el, feature_io as fio
from sfd2_amd.model import ResSegNetV2
H, W, N = 1200, 1600, 512
prec = sys.argv[1] if len(sys.argv) > 1 else "f16c"
m = ResSegNetV2(outdim=128, require_stability=True, precision=prec).eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
base = [(synth.make_image(H, W, 100 + i).transpose(1, 2, 0) * 255).astype(np.uint8).copy() for i in range(8)]
items = [{"name": f"db/{i:04d}.jpg", "image": base[i % 8], "original_size": (W, H)} for i in range(N)]
name, conf = next(iter(el.confs.items()))
class NullStore:
    path = "null"
    def close(self): pass
for label, null in (("null store", True),):
    for nw, wr, ln, dp in ((16, 3, 1, 3), (16, 3, 2, 4), (16, 3, 2, 6), (16, 3, 1, 3), (16, 3, 2, 4)):
        td = tempfile.mkdtemp()
        orig_open, orig_write = fio.open_store, fio.write_features
        if null:
            fio.open_store = lambda *a, **k: NullStore()
            fio.write_features = lambda store, nm, pred: None
        t0 = time.perf_counter()
        el.main(conf, items, td, model_and_extractor=(m, el.extract_resnet_return), num_workers=nw, writers=wr, depth=dp, lanes=ln)
        dt = time.perf_counter() - t0
        fio.open_store, fio.write_features = orig_open, orig_write
        print(f"{prec} {label}: workers {nw} writers {wr} lanes {ln} depth {dp}: {N / dt:.1f} images/s", flush=True)
