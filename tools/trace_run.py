#!/usr/bin/env python3
"""Runs 45 f16c extracts at 1600x1200 with another build of libsfd2hip bound (the -DSFD2_*_TRACE builds print their cycle stamps to stderr
after 40 launches):  python tools/trace_run.py build/variants/libpptrace.so"""
import sys, ctypes
sys.path.insert(0, "/root/repo")
from sfd2_amd import _lib
_lib.use_library(sys.argv[1])
import torch
from sfd2_amd import synth
from sfd2_amd.model import ResSegNetV2
m = ResSegNetV2(outdim=128, require_stability=True, precision="f16c").eval(); m.load_state_dict(synth.make_state_dict(0)); m.cuda(0)
ctx = m.context
H, W, K = 1200, 1600, 4096
img = torch.from_numpy(synth.make_image(H, W, 100)).cuda()
kp = torch.empty((K, 2), device="cuda"); sc = torch.empty((K,), device="cuda"); de = torch.empty((K, 128), device="cuda"); n = ctypes.c_int()
for i in range(45):
    _lib.check(ctx.lib.sfd2_extract(ctx.h, img.data_ptr(), 1, H, W, 0.001, K, _lib.FLAG_ASYNC, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), 1, K, ctypes.byref(n)))
ctx.sync()
