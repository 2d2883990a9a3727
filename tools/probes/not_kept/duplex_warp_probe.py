"""One warp_with_mask of a 4K tile on host mats: serial against banded (isx_warper_set_host_duplex), pageable and pinned buffers."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
img = synth.make_tile(H, W, 1)
warper = I.CylindricalWarper(0).create(F)
roi = warper.warpRoi((W, H), K, Rs[0])
dh, dw = roi[3] - roi[1] + 1, roi[2] - roi[0] + 1
def pinned(shape): return torch.empty(shape, dtype=torch.uint8).pin_memory().numpy()
for mem in ("pageable", "pinned"):
    mk = (lambda s: np.empty(s, np.uint8)) if mem == "pageable" else pinned
    src = mk(img.shape); src[...] = img
    di, dm = mk((dh, dw, 3)), mk((dh, dw))
    for on in (False, True, False, True):
        warper.set_host_duplex(on)
        for _ in range(3): warper.warp_with_mask(src, K, Rs[0], dst_img=di, dst_mask=dm)
        gc.disable(); t0 = time.perf_counter()
        for _ in range(20): warper.warp_with_mask(src, K, Rs[0], dst_img=di, dst_mask=dm)
        dt = (time.perf_counter() - t0) / 20; gc.enable()
        print("%-8s %-7s %.3f ms per warp (bands %d)" % (mem, "banded" if on else "serial", dt * 1e3, warper.last_host_bands()))
