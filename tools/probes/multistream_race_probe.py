"""Do P independent small pairs on S streams (bench.py's N > 1 per-rank step) always produce the serial mosaics?  (round 5: the world-8 rehearsal
saw one or two of 32 mosaics differ from their serial run, different ones every time.)
usage: multistream_race_probe.py [pairs] [streams] [steps] [W] [H] [F] [out: uint8|int16] [pitched: 0|1]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
W = int(sys.argv[4]) if len(sys.argv) > 4 else 640
H = int(sys.argv[5]) if len(sys.argv) > 5 else 360
F = float(sys.argv[6]) if len(sys.argv) > 6 else 500.0
OUT = sys.argv[7] if len(sys.argv) > 7 else "uint8"
PITCHED = int(sys.argv[8]) if len(sys.argv) > 8 else 1
dev = torch.device("cuda:0")
K, Rs = synth.camera_ring(W, H, F, 2, 0.72)
g = torch.Generator(device=dev)
streams = [None] if S <= 1 else [torch.cuda.Stream(device=dev) for _ in range(S)]
pairs, refs = [], []
for p in range(P):
    g.manual_seed(100 + p)
    imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
    ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, streams[p % len(streams)], OUT, deferred=True)
    r = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, None, OUT, deferred=True)
    refs.append(r.step_sync()[0].clone())
    pairs.append(ps)
torch.cuda.synchronize()
if PITCHED:      # bench.py's send block: rows padded to 4 bytes, all mosaics in one allocation
    shapes = [tuple(p.out.shape) for p in pairs]
    esz = pairs[0].out.element_size()
    pitches = [(sh[1] * sh[2] * esz + 3) // 4 * 4 for sh in shapes]
    n_out = sum(sh[0] * pt for sh, pt in zip(shapes, pitches))
    send = torch.empty((n_out,), dtype=torch.uint8, device=dev)
    off = 0
    for p, sh, pt in zip(pairs, shapes, pitches):
        p.out = send[off:off + sh[0] * pt].view(pairs[0].out.dtype).as_strided(sh, (pt // esz, sh[2], 1))
        off += sh[0] * pt
bad = {}
for it in range(N):
    for i, p in enumerate(pairs):
        with torch.cuda.stream(streams[i % len(streams)]) if streams[0] is not None else torch.cuda.stream(torch.cuda.current_stream()):
            p.step()
    torch.cuda.synchronize()
    for i, p in enumerate(pairs):
        if not torch.equal(p.out, refs[i]):
            d = (p.out != refs[i]).any(dim=2)
            ys, xs = torch.nonzero(d, as_tuple=True)
            bad.setdefault(i, []).append((it, int(d.sum()), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())))
for p in pairs:
    p.check_plan()
print("pairs %d streams %d steps %d %dx%d out %s pitched %d: mismatching (pair: [(step, px, x0, x1, y0, y1) ...])" % (P, S, N, W, H, OUT, PITCHED),
      {k: v[:4] for k, v in bad.items()}, "total", sum(len(v) for v in bad.values()))
