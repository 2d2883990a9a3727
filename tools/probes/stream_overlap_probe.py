"""Do two HIP streams really run kernels concurrently on this box?  A chain of tiny (latency-bound) kernels on one stream, a few large
(bandwidth-bound) kernels on another: alone, then together.  python tools/probes/stream_overlap_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
small = torch.zeros(4096, device=dev)
big = torch.zeros(64 * 1024 * 1024, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def chain_small(n):
    with torch.cuda.stream(sa):
        for _ in range(n):
            small.add_(1.0)


def chain_big(n):
    with torch.cuda.stream(sb):
        for _ in range(n):
            big.mul_(1.0001)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    chain_small(50); chain_big(5)
torch.cuda.synchronize()
NS, NB = 400, 20
ta = timed(lambda: chain_small(NS))
tb = timed(lambda: chain_big(NB))
tab = timed(lambda: (chain_big(NB), chain_small(NS)))
tba = timed(lambda: (chain_small(NS), chain_big(NB)))
print("small chain alone %.3f ms (%d kernels, %.2f us each); big alone %.3f ms (%d kernels, %.1f us each)" % (ta, NS, ta / NS * 1e3, tb, NB, tb / NB * 1e3))
print("both streams: %.3f ms (big enqueued first), %.3f ms (small enqueued first); serial sum %.3f ms, perfect overlap %.3f ms" % (tab, tba, ta + tb, max(ta, tb)))
