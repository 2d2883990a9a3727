"""How long is a fenced region of K planned steps?  t(K) = a + b K: b is the steady per-step time, a the fixed cost of the fences and
of the pipeline filling and draining (it is what separates a 20-step bench line from a 200-step one).  Run on the GPU box."""
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imagestitch_amd
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher

W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
lib = imagestitch_amd.load()
p = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16")
for _ in range(5):
    p.step()
torch.cuda.synchronize()
gc.collect(); gc.disable()


def region(k, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, th * 1e3


def fit(fn, label):
    ks = [1, 2, 5, 10, 20, 50, 100]
    ts = []
    for k in ks:
        ts.append(min(region(k, fn)[0] for _ in range(5)))
    A = np.vstack([np.ones(len(ks)), ks]).T
    a, b = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
    print("%-44s a = %.3f ms, b = %.4f ms/step; t(1) = %.3f, t(20)/20 = %.4f" % (label, a, b, ts[0], ts[4] / 20))


fit(p.step, "planned step")
if hasattr(p.warper, "set_roi_cache"):
    p.warper.set_roi_cache(True)
    fit(p.step, "planned step, ROI cache (no verify scans)")
    p.warper.set_roi_cache(False)
lib.isx_profile_enable(1); lib.isx_profile_filter(b"collapse_roll"); lib.isx_profile_sample(4)
fit(p.step, "planned step, dominant kernel bracketed 1/4")
lib.isx_profile_enable(0)
p.capture()
fit(p.replay, "hipGraph replay")

# a bench-like sequence: idle for `pause` ms, W warm steps, fence, ONE region of 20 steps
p2 = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16")
for pause in (0, 5, 40, 200):
    for wsteps in (0, 3, 20):
        r = []
        for _ in range(4):
            torch.cuda.synchronize(); time.sleep(pause * 1e-3)
            for _ in range(wsteps):
                p2.step()
            r.append(region(20, p2.step)[0] / 20)
        print("idle %3d ms, %2d warm steps, then 20 timed: %s ms per step" % (pause, wsteps, " ".join("%.4f" % x for x in r)))
