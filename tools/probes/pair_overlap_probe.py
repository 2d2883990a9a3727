"""Two independent 4K pairs on two streams: do their steps overlap?  (run on the GPU box)  Modes: eager on one stream, eager on two
streams, one hipGraph per pair on its own stream.  Reports ms per pair and the host's enqueue time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher

W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
streams = [torch.cuda.Stream() for _ in range(NP)]
pairs = []
for p in range(NP):
    imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
    pairs.append(PairStitcher(imgs, K, Rs, F, "cylindrical", 5, _lib.PREC_F32, 0, streams[p], "int16"))
torch.cuda.synchronize()


def run(n):
    for _ in range(n):
        for p, s in zip(pairs, streams):
            with torch.cuda.stream(s):
                p.step()


run(3); torch.cuda.synchronize()
n = 30
t0 = time.perf_counter(); run(n); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d pairs, eager, one stream each: %.3f ms per pair (host enqueue %.3f ms per pair)" % (NP, dt / n / NP * 1e3, th / n / NP * 1e3))
for p in pairs:
    p.capture()
torch.cuda.synchronize()
for _ in range(3):
    for p in pairs:
        p.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    for p in pairs:
        p.replay()
th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d pairs, one hipGraph each on its own stream: %.3f ms per pair (host %.3f ms per pair)" % (NP, dt / n / NP * 1e3, th / n / NP * 1e3))
