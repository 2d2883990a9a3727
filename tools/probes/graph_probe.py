"""hipGraph replay of the planned step against the eager step (one 4K pair, fp32): alternating, several repetitions."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagestitch_amd import synth
from imagestitch_amd.pipeline import PairStitcher
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
def bench(step, n=60):
    for _ in range(5): step()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    gc.enable()
    return dt
pe = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, 1, 0, None, "int16")
pg = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, 1, 0, None, "int16")
pg.capture()
for rep in range(5):
    te, tg = bench(pe.step), bench(pg.replay)
    print("eager %.4f ms %.1f Gpix/s | graph %.4f ms %.1f Gpix/s | graph / eager %.3f" % (te, 2 * W * H / te / 1e6, tg, 2 * W * H / tg / 1e6, te / tg))
print("plan", pe.check_plan(), pg.check_plan())
