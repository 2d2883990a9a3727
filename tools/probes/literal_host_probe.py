"""Host time per call of the literal drop-in sequence (step_literal) on a 4K pair: where the caller's thread spends a step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth, _lib as L
from imagestitch_amd.blender import convert_to
from imagestitch_amd.pipeline import PairStitcher

W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, int(sys.argv[1]) if len(sys.argv) > 1 else 1, 0, None, "int16", deferred="copy")
for _ in range(3):
    ps.step_literal()
torch.cuda.synchronize()
acc = {}
def t(name, fn):
    t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0; return r
N = 20
t_all = time.perf_counter()
for _ in range(N):
    cs = []
    for i in range(2):
        im = ps.imgs[i]; size = (W, H)
        roi = t("roi", lambda: ps.warper.warpRoi(size, K, Rs[i]))
        t("warp_img", lambda: ps.warper.warp_roi(im, K, Rs[i], L.INTER_LINEAR, L.BORDER_REFLECT, roi, ps.lit_warped[i]))
        roi = t("roi", lambda: ps.warper.warpRoi(size, K, Rs[i]))
        t("warp_mask", lambda: ps.warper.warp_roi(ps.src_masks[i], K, Rs[i], L.INTER_NEAREST, L.BORDER_CONSTANT, roi, ps.lit_wmasks[i]))
        cs.append((roi[0], roi[1]))
        t("convert", lambda: convert_to(ps.lit_warped[i], np.int16, dst=ps.warped16[i]))
    t("prepare", lambda: ps.blender.prepare(cs, ps.sizes))
    for i in range(2):
        t("feed", lambda: ps.blender.feed(ps.warped16[i], ps.seam[i], cs[i]))
    t("blend", lambda: ps.blender.blend(ps.lit_out, ps.lit_out_mask))
host = (time.perf_counter() - t_all) / N
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N
print("host ms/step %.3f  wall ms/step %.3f" % (host * 1e3, wall * 1e3))
for k, v in acc.items():
    print("  %-10s %7.1f us per step" % (k, v / N * 1e6))
