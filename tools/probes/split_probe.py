"""One 4K pair as nsplit staggered column strips on nsplit streams (pipeline.SplitStitcher) against the single chain (PairStitcher):
identical mosaics, ms per step of each.  usage: split_probe.py [prec] [nsplit ...]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from imagestitch_amd import synth
from imagestitch_amd.pipeline import PairStitcher, SplitStitcher

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
splits = [int(v) for v in sys.argv[2:]] or [2]
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]

def bench(step, n=40):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    gc.enable()
    return dt

ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, None, "int16")
ref, refm = [t.clone() for t in ps.step()]
print("single chain        %.4f ms per step  %.1f Gpix/s" % ((lambda t: (t, 2 * W * H / t / 1e6))(bench(ps.step))))
for ns in splits:
    for stag in (1, 0, None):
        for chain in (False, True):
            sp = SplitStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, "int16", nsplit=ns, stagger_level=stag, chain_steps=chain)
            out, m = sp.step()
            torch.cuda.synchronize()
            same = bool(torch.equal(out, ref) and torch.equal(m, refm))
            t = bench(sp.step)
            print("split %d stagger %-4s chain_steps %-5s identical %s  %.4f ms per step  %.1f Gpix/s  plan %d" % (ns, stag, chain, same, t, 2 * W * H / t / 1e6, sp.check_plan()))
            del sp
