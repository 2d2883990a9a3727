"""SURVEY N3's fusions timed on a 4K pair of device tiles (W:238-313 without the seam finder): gain apply + mask preparation as passes of
their own (isx_gain_apply, isx_mask_dilate_and, isx_blender_feed) against folded into the warp's store and into the feed
(isx_warper_set_gain, isx_blender_feed_dilated).  Same mosaics; ms per pair and the launches of each form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth, _lib

W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
gains = [0.98872, 1.0114]
warper = I.CylindricalWarper().create(F)
rois = [warper.warpRoi((W, H), K, R) for R in Rs]
sizes = [(r[2] - r[0] + 1, r[3] - r[1] + 1) for r in rois]
corners = [(r[0], r[1]) for r in rois]
def pitched(h, w, c, dt=torch.uint8):
    p = (w * c + 63) // 64 * 64
    t = torch.empty((h * p,), dtype=dt, device=dev)
    return t.as_strided((h, w, c), (p, c, 1)) if c > 1 else t.as_strided((h, w), (p, 1))
warped = [pitched(h, w, 3) for (w, h) in sizes]
wmasks = [pitched(h, w, 1) for (w, h) in sizes]
for i in range(2):
    warper.warp_with_mask(imgs[i], K, Rs[i], dst_img=warped[i], dst_mask=wmasks[i])
seam = [torch.from_numpy(s).to(dev) for s in synth.seam_masks(corners, [m.cpu().numpy() for m in wmasks])]   # the seam finder's output (not timed here)
mk = [torch.empty_like(s) for s in seam]
outs = {}
for kind, make in (("multiband f32", lambda: I.MultiBandBlender(False, 5, I.PREC_F32)), ("feather 0.1", lambda: I.FeatherBlender(False, 0.1))):
    for fused in (False, True):
        b = make()
        b.set_deferred_level0("copy")
        def step():
            for i in range(2):
                if fused:
                    warper.set_gain(gains[i])
                    warper.warp_with_mask(imgs[i], K, Rs[i], dst_img=warped[i], dst_mask=wmasks[i])
                else:
                    warper.warp_with_mask(imgs[i], K, Rs[i], dst_img=warped[i], dst_mask=wmasks[i])
                    I.gain_apply(warped[i], gains[i])
            warper.set_gain(1.0)
            b.prepare(corners, sizes)
            for i in range(2):
                if fused:
                    b.feed_dilated(warped[i], seam[i], wmasks[i], 20, 20, corners[i])
                else:
                    I.blender.dilate_and(seam[i], 20, 20, other=wmasks[i]) if False else None
                    mm, mo, mt = _lib.as_mat(seam[i]), _lib.as_mat(mk[i]), _lib.as_mat(wmasks[i])
                    import ctypes as C
                    _lib.check(_lib.load().isx_mask_dilate_and(C.byref(mm), C.byref(mt), 20, 20, C.byref(mo), 0, None))
                    (b.feed_u8 if kind.startswith("multi") else b.feed_u8)(warped[i], mk[i], corners[i])
            return b.blend()
        for _ in range(3):
            d, m = step()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            d, m = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        lib = _lib.load()
        lib.isx_profile_enable(1); lib.isx_profile_reset()
        d, m = step()
        ent = _lib.profile_entries()
        lib.isx_profile_enable(0)
        outs[(kind, fused)] = (d.clone(), m.clone())
        launches = sum(v["launches"] for v in ent.values())
        print("%-14s %-8s %.4f ms per pair, %d launches: %s" % (kind, "fused" if fused else "separate", dt * 1e3, launches,
              ", ".join("%s %.1f us" % (k, v["ms"] * 1e3) for k, v in sorted(ent.items(), key=lambda kv: -kv[1]["ms"]) if k in ("gain_apply", "dilate_and", "feed_copy", "warp_tile"))))
    a, b2 = outs[(kind, False)], outs[(kind, True)]
    print("%-14s identical mosaics: %s" % (kind, bool(torch.equal(a[0], b2[0]) and torch.equal(a[1], b2[1]))))
