"""Does the step time depend on how long the GPU has been busy?  One 4K pair, fp32, the planned eager step: after the set-up
(GPU idle), blocks of 5 steps timed back to back for 60 blocks; then the same after 0.5 s of host sleep."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagestitch_amd import synth
from imagestitch_amd.pipeline import PairStitcher
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
p = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, 1, 0, None, "int16")
p.step(); torch.cuda.synchronize()
gc.collect(); gc.disable()
def blocks(nb, per=5):
    out = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(nb):
        for _ in range(per): p.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter(); out.append((t1 - t0) / per * 1e3); t0 = t1
    return out
for rep in range(3):
    time.sleep(0.5)
    ts = blocks(60)
    print("after 0.5 s idle, ms per step in blocks of 5:", " ".join("%.3f" % t for t in ts))
    print("  first 5 steps %.4f | steps 5-25 %.4f (what --warmup 5 --steps 20 times) | steps 100-300 %.4f" % (ts[0], sum(ts[1:5]) / 4, sum(ts[20:]) / 40))
