"""Two independent 4K pairs on two streams: do their steps overlap?  (run on the GPU box)  Modes: eager on one stream, eager on two
streams, one hipGraph per pair on its own stream.  Reports ms per pair and the host's enqueue time per step."""
import gc
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
NS = int(sys.argv[2]) if len(sys.argv) > 2 else NP
_pool = [torch.cuda.Stream() for _ in range(NS)]
streams = [_pool[p % NS] for p in range(NP)]
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
gc.collect(); gc.disable()   # a generation-2 collection (~35 ms) inside a timed loop looks like a stalled GPU
n = 30
t0 = time.perf_counter(); run(n); th = time.perf_counter() - t0; torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%d pairs, eager, %d streams: %.3f ms per pair (host enqueue %.3f ms per pair)" % (NP, NS, dt / n / NP * 1e3, th / n / NP * 1e3))
if len(sys.argv) > 3:
    # per-call enqueue times: a flat distribution = the host's own cost, rare long calls = the runtime blocking on the GPU
    import numpy as np
    ts = []
    for _ in range(10):
        for p, s in zip(pairs, streams):
            with torch.cuda.stream(s):
                a = time.perf_counter(); p.step(); ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    ts = np.array(ts) * 1e3
    print("  step() enqueue ms: min %.3f median %.3f p90 %.3f max %.3f; calls > 2x median: %d of %d" % (
        ts.min(), np.median(ts), np.percentile(ts, 90), ts.max(), int((ts > 2 * np.median(ts)).sum()), ts.size))
    # the same with the GPU idle between pairs: the host's unblocked cost
    ts2 = []
    for p, s in zip(pairs, streams):
        with torch.cuda.stream(s):
            a = time.perf_counter(); p.step(); ts2.append(time.perf_counter() - a)
        torch.cuda.synchronize()
    print("  step() enqueue ms with the GPU idle: median %.3f" % (np.median(ts2) * 1e3))
    # the host kept at most DEPTH whole iterations ahead of the GPU (an event per iteration, waited on DEPTH iterations later)
    for depth in (1, 2, 4):
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(n):
            for p, s in zip(pairs, streams):
                with torch.cuda.stream(s):
                    p.step()
            row = []
            for s in _pool:
                e = torch.cuda.Event(); e.record(s); row.append(e)
            evs.append(row)
            if it >= depth:
                for e in evs[it - depth]:
                    e.synchronize()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("  host at most %d iterations ahead: %.3f ms per pair" % (depth, dt / n / NP * 1e3))
    sys.exit(0)
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
