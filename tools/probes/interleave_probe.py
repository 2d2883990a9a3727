"""PairStitcher with interleave (warp t, feed t with its Gaussian chain on a side stream, warp t+1 ...) vs the default order."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
for inter in (False, True):
    ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, "int16", interleave=inter)
    for _ in range(5): ps.step()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): ps.step()
    torch.cuda.synchronize()
    print("interleave=%s: %.4f ms per pair" % (inter, (time.perf_counter() - t0) / n * 1e3))
    ps.check_plan()
    del ps
