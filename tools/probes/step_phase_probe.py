"""Where a synchronous step (the reference's call sequence: corner back to the host per warp) spends its time: host time and
synchronised time per call.  python tools/probes/step_phase_probe.py [f32|i16|f16acc32] [deferred|copy|eager]   (4K pair, cylindrical)"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imagestitch_amd
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher

prec = {"f32": _lib.PREC_F32, "i16": _lib.PREC_I16, "f16acc32": _lib.PREC_F16ACC32}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
cycle = {"deferred": True, "copy": "copy", "eager": False}[sys.argv[2] if len(sys.argv) > 2 else "copy"]
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
p = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, None, "int16", deferred=cycle)
for _ in range(5):
    p.step_sync()
torch.cuda.synchronize()
gc.collect(); gc.disable()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    p.step_sync()
th = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ph = {}


def T(name, fn):
    torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    ph[name] = ph.get(name, 0) + (c - a); ph[name + "_host"] = ph.get(name + "_host", 0) + (b - a)
    return r


R = 20
for _ in range(R):
    cs = []
    for i in range(2):
        cs.append(T("warp", lambda: p.warper.warp_with_mask(p.imgs[i], p.K, p.Rs[i], dst_img=p.warped[i], dst_mask=p.wmasks[i]))[0])
    T("prepare", lambda: p.blender.prepare(cs, p.sizes))
    for i in range(2):
        T("feed", lambda: p.blender.feed_u8(p.warped[i], p.seam[i], cs[i]))
    T("blend", lambda: p.blender.blend(p.out, p.out_mask))
print(sys.argv[1:], "step_sync %.4f ms (host %.4f)" % (dt / n * 1e3, th / n * 1e3), {k: round(v / R * 1e3, 4) for k, v in ph.items()})
