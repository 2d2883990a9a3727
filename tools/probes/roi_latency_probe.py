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
print("idle GPU: %.1f us per call" % run())
a = torch.empty((1 << 28,), dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(400):
        a.add_(1)
print("busy GPU: %.1f us per call" % run())
torch.cuda.synchronize()
