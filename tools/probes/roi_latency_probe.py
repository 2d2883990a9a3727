"""Host time of one synchronous detectResultRoi (isx_warper_roi) on a 4K source: idle GPU, and with the GPU busy on another stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
w = I.CylindricalWarper().create(F)
for _ in range(20):
    w.warpRoi((W, H), K, Rs[0])
def run(n=300):
    t0 = time.perf_counter()
    for i in range(n):
        w.warpRoi((W, H), K, Rs[i & 1])
    return (time.perf_counter() - t0) / n * 1e6
# the C call alone (arguments converted once: the wrapper's numpy -> ctypes conversions are the caller's, 4 - 5 us of the figure above)
import ctypes as C
from imagestitch_amd import _lib
lib = _lib.load()
_k, kp = _lib.f9(K)
_rs = [_lib.f9(R) for R in Rs]
roi4 = (C.c_int * 4)()
def run_c(n=300):
    t0 = time.perf_counter()
    for i in range(n):
        lib.isx_warper_roi(w._h, W, H, kp, _rs[i & 1][1], roi4, None)
    return (time.perf_counter() - t0) / n * 1e6
print("idle GPU: %.1f us per call (C call alone: %.1f)" % (run(), run_c()))
a = torch.empty((1 << 28,), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(400):
        a.add_(1)
print("busy GPU: %.1f us per call (C call alone: %.1f)" % (run(), run_c()))
torch.cuda.synchronize()
