"""Per-launch time of the fused tile warp on one 4K tile (run on the GPU box): python tools/warp_probe.py [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagestitch_amd as I
from imagestitch_amd import _lib, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
img = torch.from_numpy(synth.make_tile(H, W, 0)).to(dev)
w = I.CylindricalWarper().create(F)
roi = w.warpRoi((W, H), K, Rs[0])
dw, dh = roi[2] - roi[0] + 1, roi[3] - roi[1] + 1
pi = torch.empty((dh * ((dw * 3 + 63) // 64 * 64),), dtype=torch.uint8, device=dev).as_strided((dh, dw, 3), ((dw * 3 + 63) // 64 * 64, 3, 1))
pm = torch.empty((dh * ((dw + 63) // 64 * 64),), dtype=torch.uint8, device=dev).as_strided((dh, dw), ((dw + 63) // 64 * 64, 1))
lib = _lib.load()
mi, mdi, mdm = _lib.as_mat(img), _lib.as_mat(pi), _lib.as_mat(pm)
_k, kp = _lib.f9(K); _r, rp = _lib.f9(Rs[0])
r4 = (C.c_int * 4)(*roi)
def call():
    _lib.check(lib.isx_warper_warp_with_mask_roi(w._h, C.byref(mi), None, kp, rp, r4, C.byref(mdi), C.byref(mdm)))
for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
print("warp %dx%d -> %dx%d: %.2f us per launch (dbg=%s rows=%s v1=%s)" % (W, H, dw, dh, e0.elapsed_time(e1) / reps * 1e3, os.environ.get("ISX_WARP_DBG"), os.environ.get("ISX_WARP_ROWS"), os.environ.get("ISX_WARP_V1")))
