"""Development probe: isx_seam_estimate on the overlap of a 4K-sized pair (device-resident inputs) vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import _lib
from oracle import capi as O
from seam_cases import make_case

c = make_case(7, size1=(2169, 3417), size2=(2169, 3417), tl1=(-1709, -1085), tl2=(451, -1085), holes=True)
print("roi", c["roi"], "p1", c["p1"], "p2", c["p2"])
t0 = time.time(); ref, _ = O.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"]); tc = time.time() - t0
d = {k: torch.from_numpy(c[k]).cuda() for k in ("img1", "img2", "labels")}
lib = _lib.load()
for _ in range(2):
    got, _ = I.seam_estimate(d["img1"], d["img2"], c["tl1"], c["tl2"], c["union_tl"], d["labels"], c["label"], c["roi"], c["p1"], c["p2"])
assert np.array_equal(got, ref), "mismatch"
lib.isx_profile_enable(1); lib.isx_profile_reset()
t0 = time.time(); n = 5
for _ in range(n):
    I.seam_estimate(d["img1"], d["img2"], c["tl1"], c["tl2"], c["union_tl"], d["labels"], c["label"], c["roi"], c["p1"], c["p2"])
tg = (time.time() - t0) / n
ent = _lib.profile_entries()
print("seam points", len(ref), " CPU oracle %.1f ms   GPU call %.2f ms" % (tc * 1e3, tg * 1e3), {k: round(v["ms"] / n, 3) for k, v in ent.items()})
