"""Can the NEXT step's warps run under THIS step's blend?  (round 6)

The pair's step is the sum of its kernels (no gaps), and 42 us of its 175 are the small-level tail: launches that leave most of the chip idle
while they walk their dependent phases.  Inside one step nothing independent is left to run beside them - but the warps of the FOLLOWING step are
(given a second set of warped-tile buffers: the last collapse step reads the tiles).  This probe measures what that overlap is worth before
anything is built on it:  stitcher A's blend chain (prepare, feeds, blend) on one stream, stitcher B's two planned warps on another.

    python tools/probes/warp_under_blend_probe.py [steps]

Prints ms per iteration for: blend alone, warps alone, both enqueued per iteration (A's blend first / B's warps first), and the whole step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagestitch_amd import synth, _lib as L
from imagestitch_amd.pipeline import PairStitcher

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
pa = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, L.PREC_F32, 0, sa, "int16")
pb = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, L.PREC_F32, 0, sb, "int16")
for p in (pa, pb):
    for _ in range(3):
        p.step()
torch.cuda.synchronize()


def warps(p):
    for i in p.active:
        p.warper.warp_with_mask_planned(p.imgs[i], p.K, p.Rs[i], p.rois[i], p.warped[i], p.wmasks[i])
    p.warper.verify()


def blend(p):
    p.blender.prepare(p.corners, p.sizes)
    for i in p.active:
        p._feed(i, p.corners[i])
    p.blender.blend(p.out, p.out_mask)


def timed(fn, n=N):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rep in range(3):
    t_step = timed(lambda: pa.step())
    t_blend = timed(lambda: blend(pa))
    t_warp = timed(lambda: warps(pb))
    t_ab = timed(lambda: (blend(pa), warps(pb)))
    t_ba = timed(lambda: (warps(pb), blend(pa)))
    print("step %.4f | blend alone %.4f  warps alone %.4f  sum %.4f | both, blend enqueued first %.4f  warps first %.4f ms per iteration" % (
        t_step, t_blend, t_warp, t_blend + t_warp, t_ab, t_ba), flush=True)
pa.check_plan(); pb.check_plan()
