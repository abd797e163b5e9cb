"""Race screen for the hand-synchronised kernels (counted vmcnt, LDS-only barriers, persistent blocks): the same
extract + match step repeated N times -- alone and with a second context hammering the GPU to perturb timing -- must
give bit-identical outputs every time.  A read placed one barrier too early shows up here as a rare differing run."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
PREC = sys.argv[2] if len(sys.argv) > 2 else "f16c"      # python tools/determinism_check.py 1000 [f16c | f16 | f16x3 | f32]
dev = torch.device("cuda", 0)
sd = synth.make_state_dict(0)


def lane(seed, H, W, K, KDB):
    m = ResSegNetV2(outdim=128, require_stability=True, precision=PREC).eval()
    m.load_state_dict(sd)
    m.cuda(0)
    img = torch.from_numpy(synth.make_image(H, W, seed)).to(dev)
    db = [torch.from_numpy(synth.make_descriptors(K, seed=seed + 1 + i)).to(torch.float16).to(dev).contiguous() for i in range(KDB)]
    dbs = (_lib.DescSet * KDB)(*[_lib.DescSet(d.data_ptr(), K, _lib.DT_F16, _lib.LAYOUT_ND, 1) for d in db])
    out = dict(kp=torch.zeros((K, 2), device=dev), sc=torch.zeros((K,), device=dev), de=torch.zeros((K, 128), device=dev),
               mt=torch.zeros((KDB, K), dtype=torch.int64, device=dev), ms=torch.zeros((KDB, K), device=dev))
    q = _lib.DescSet(out["de"].data_ptr(), K, _lib.DT_F32, _lib.LAYOUT_ND, 1)
    mconf = _lib.MatchConf(_lib.MATCH_HLOC, 1, 0.0, 0.0, _lib.SIM_F16)
    n = ctypes.c_int()
    ctx = m.context

    def step():
        _lib.check(ctx.lib.sfd2_extract(ctx.h, img.data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, out["kp"].data_ptr(),
                                        out["sc"].data_ptr(), out["de"].data_ptr(), 1, K, ctypes.byref(n)))
        _lib.check(ctx.lib.sfd2_match_batch(ctx.h, ctypes.byref(q), dbs, KDB, 128, ctypes.byref(mconf), out["mt"].data_ptr(),
                                            out["ms"].data_ptr(), 1, _lib.FLAG_ASYNC))
    return m, ctx, step, out, (img, db, dbs, q, mconf, n)


bad = 0
for (H, W, K, KDB) in [(1200, 1600, 4096, 8), (1024, 1024, 4096, 4), (477, 635, 2048, 3)]:
    m, ctx, step, out, keep = lane(7, H, W, K, KDB)
    m2, ctx2, step2, out2, keep2 = lane(99, 600, 800, 1024, 2)
    step(); ctx.sync()
    ref = {k: v.clone() for k, v in out.items()}
    for phase in ("alone", "with a second context running"):
        diffs = 0
        for i in range(N):
            if phase != "alone":
                step2()
            step()
            if phase != "alone":
                step2()
            ctx.sync()
            if any(not torch.equal(out[k], ref[k]) for k in ref):
                diffs += 1
        ctx2.sync()
        print(f"[{PREC}] {W}x{H} top-{K}, {KDB} db sets, {N} runs {phase}: {diffs} differing")
        bad += diffs
print("DETERMINISTIC" if bad == 0 else f"NON-DETERMINISTIC: {bad}")
sys.exit(1 if bad else 0)
