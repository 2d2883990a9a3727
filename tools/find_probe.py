"""Development probe: isx_dp_seam_find (host component logic + GPU estimateSeam) on a 4K-sized pair vs the Python oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import _lib
from oracle.dpseam_np import DpSeamFinder as OracleFinder
from seam_cases import make_find_case

images, corners, masks = make_find_case(3, 2, False, holes=False, size=(2169, 3417))
print("tiles", [im.shape for im in images], "corners", corners)
ref = [m.copy() for m in masks]
t0 = time.time(); OracleFinder().find(images, corners, ref); tc = time.time() - t0
dimg = [torch.from_numpy(im).cuda() for im in images]
lib = _lib.load()
for _ in range(2):
    got = [m.copy() for m in masks]
    I.DpSeamFinder().find(dimg, corners, got)
assert all(np.array_equal(a, b) for a, b in zip(got, ref)), "mismatch"
lib.isx_profile_enable(1); lib.isx_profile_reset()
n = 5
work = [[m.copy() for m in masks] for _ in range(n)]        # find() edits the masks in place: fresh copies, made outside the timed region
t0 = time.time()
for got in work:
    I.DpSeamFinder().find(dimg, corners, got)
tg = (time.time() - t0) / n
assert all(np.array_equal(a, b) for a, b in zip(work[-1], ref)), "mismatch"
ent = _lib.profile_entries()
print("Python oracle %.0f ms   isx_dp_seam_find %.1f ms per call (GPU kernels: %s)" % (tc * 1e3, tg * 1e3, {k: round(v["ms"] / n, 3) for k, v in ent.items()}))
