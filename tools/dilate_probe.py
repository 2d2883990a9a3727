"""Development probe: isx_mask_dilate_and on a 4K-tile-sized device mask for several structuring elements."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagestitch_amd as I
from imagestitch_amd import _lib

dev = torch.device("cuda:0")
h, w = 2169, 3417
m = torch.full((h, w), 255, dtype=torch.uint8, device=dev); m[:40, :300] = 0
lib = _lib.load()
for k in [(1, 1), (3, 3), (20, 20), (33, 33), (34, 34)]:
    for with_other in (False, True):
        for _ in range(3): I.dilate_and(m, *k, other=m if with_other else None)
        torch.cuda.synchronize()
        lib.isx_profile_enable(1); lib.isx_profile_reset()
        n = 10
        for _ in range(n): I.dilate_and(m, *k, other=m if with_other else None)
        ent = _lib.profile_entries()
        lib.isx_profile_enable(0)
        print(k, "other" if with_other else "     ", {a: round(v["ms"] / n * 1e3, 1) for a, v in ent.items()}, "us")
import time
s = torch.cuda.Stream()
for k in [(1, 1), (20, 20)]:
    for st in (None, s):
        out = None
        torch.cuda.synchronize()
        t0 = time.time(); n = 200
        for _ in range(n): out = I.dilate_and(m, *k, other=m, stream=st)
        th = time.time() - t0
        torch.cuda.synchronize()
        print(k, "stream" if st else "null  ", "wall us/call %.1f (host enqueue %.1f)" % ((time.time() - t0) / n * 1e6, th / n * 1e6))
