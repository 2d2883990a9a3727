"""PairStitcher(interleave=True) - each tile's Gaussian chain on a side stream under the next tile's warp - against the plain chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagestitch_amd import synth
from imagestitch_amd.pipeline import PairStitcher
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
def bench(step, n=40):
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for il in (False, True):
        ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, 1, 0, None, "int16", interleave=il)
        t = bench(ps.step)
        print("interleave %-5s %.4f ms per step %.1f Gpix/s" % (il, t, 2 * W * H / t / 1e6))
        del ps
