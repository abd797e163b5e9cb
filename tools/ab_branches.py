import sys, ctypes, time
sys.path.insert(0,'/root/repo')
import torch, numpy as np
from sfd2_amd import _lib, synth
from sfd2_amd.model import ResSegNetV2
sd = synth.make_state_dict(0)
m = ResSegNetV2(outdim=128, require_stability=True, precision="f16").eval(); m.load_state_dict(sd); m.cuda(0)
ctx = m.context; lib = ctx.lib
H,W,K = 1200,1600,4096
imgs=[torch.from_numpy(synth.make_image(H,W,100+i)).cuda() for i in range(4)]
kp=torch.empty((K,2),device='cuda'); sc=torch.empty((K,),device='cuda'); de=torch.empty((K,128),device='cuda'); n=ctypes.c_int()
def run(reps):
    for i in range(reps):
        _lib.check(lib.sfd2_extract(ctx.h, imgs[i%4].data_ptr(),1,H,W,0.001,K,_lib.FLAG_ASYNC,kp.data_ptr(),sc.data_ptr(),de.data_ptr(),1,K,ctypes.byref(n)))
    ctx.sync()
res={}
for rnd in range(3):
    for br in (0,1):
        ctx.set_option("branches", br)
        run(20)
        t0=time.perf_counter(); run(100); dt=(time.perf_counter()-t0)/100
        res.setdefault(br,[]).append(dt*1e3)
        if rnd==0:
            out=(kp.clone(),sc.clone(),de.clone())
            if br==0: ref=out
            else: print("bit-identical outputs:", all(torch.equal(a,b) for a,b in zip(ref,out)))
print({k:[round(x,4) for x in v] for k,v in res.items()})
