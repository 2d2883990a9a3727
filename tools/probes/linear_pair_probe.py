"""A13 (the reference's in-tree linear-ramp pair blend, B:141-717) at the size of a warped 4K pair: time per call on device mats, per
kernel (run on the GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imagestitch_amd
from imagestitch_amd import _lib

rng = np.random.default_rng(41)
h1, w1, h2, w2 = 2170, 3416, 2166, 3416
yy, xx = np.mgrid[0:h1, 0:w1].astype(np.float32)
base = 120.0 + 60.0 * np.sin(xx / 97.0) * np.cos(yy / 61.0)
img1 = np.clip(base[..., None] + rng.normal(0, 1, (h1, w1, 3)), 0, 255).astype(np.float32)
shifted = np.roll(base, -2159, axis=1)[:h2, :w2]
valley = np.minimum(0.8 * np.abs(xx[:h2, :w2] - (628.0 + 0.25 * yy[:h2, :w2])), 60.0)
valley[:, 1257:] = 0
img2 = np.clip((shifted + valley)[..., None] + rng.normal(0, 1, (h2, w2, 3)), 0, 255).astype(np.float32)
tl1, tl2 = (-2788, -1085), (-2788 + 2159, -1085 + 4)
lib = imagestitch_amd.load()
t1, t2 = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
ph = max(tl1[1] + h1, tl2[1] + h2) - min(tl1[1], tl2[1])
pw = max(tl1[0] + w1, tl2[0] + w2) - min(tl1[0], tl2[0])
m1, m2 = _lib.as_mat(t1), _lib.as_mat(t2)
# the panorama's size as the library reports it
pano = None
for shape in ((ph, pw, 3), (ph + 1, pw + 1, 3), (ph, pw + 1, 3), (ph + 1, pw, 3)):
    cand = torch.empty(shape, dtype=torch.float32, device="cuda")
    seam = np.zeros(shape[0], np.int32)
    mp = _lib.as_mat(cand)
    if lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), tl1[0], tl1[1], tl2[0], tl2[1], C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None) == 0:
        pano = cand
        break
assert pano is not None, lib.isx_last_error()
mp = _lib.as_mat(pano)
seam = np.zeros(pano.shape[0], np.int32)


def call():
    _lib.check(lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), tl1[0], tl1[1], tl2[0], tl2[1], C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None))


for _ in range(3):
    call()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
lib.isx_profile_enable(1); lib.isx_profile_filter(None); lib.isx_profile_reset()
call()
torch.cuda.synchronize()
ent = _lib.profile_entries()
lib.isx_profile_enable(0)
print("A13 on two %dx%d CV_32FC3 tiles (overlap 1257 columns) -> %dx%d panorama: %.3f ms per call (the seam goes back to the host: one synchronisation)" % (w1, h1, pano.shape[1], pano.shape[0], dt * 1e3))
print({k: round(v["ms"], 4) for k, v in ent.items()})
