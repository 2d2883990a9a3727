import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import imagestitch_amd
from imagestitch_amd import synth, _lib, mosaic
from imagestitch_amd.pipeline import MosaicStitcher
W, H, F, NT = 7680, 4320, 6000.0, 8
K, Rs = synth.camera_ring(W, H, F, NT, 0.55)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(NT)]
for window in (None, (11520, 15360)):
    for cache in (False, True):
        p = MosaicStitcher([im if (window is None or i in (2, 3, 4)) else None for i, im in enumerate(imgs)], K, Rs, F, "spherical", 7, _lib.PREC_F16ACC32, 0, None, "int16", window=window)
        p.warper.set_roi_cache(cache)
        for _ in range(3):
            p.step_sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            p.step_sync()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # phases, synchronised
        ph = {}
        def T(name, fn):
            torch.cuda.synchronize(); a = time.perf_counter(); r = fn(); b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
            ph[name] = ph.get(name, 0) + (c - a); ph[name + "_host"] = ph.get(name + "_host", 0) + (b - a)
            return r
        cs = list(p.corners)
        for i in p.active:
            if p.tile_cols is not None:
                p.warper.set_dst_columns(*p.tile_cols[i])
            cs[i] = T("warp", lambda: p.warper.warp_with_mask(p.imgs[i], p.K, p.Rs[i], dst_img=p.warped[i], dst_mask=p.wmasks[i]))[0]
        p.warper.set_dst_columns(0, 0)
        T("prepare", lambda: p.blender.prepare(cs, p.sizes))
        for i in p.active:
            T("feed", lambda: p.blender.feed_u8(p.warped[i], p.seam[i], cs[i]))
        T("blend", lambda: p.blender.blend(p.out, p.out_mask))
        print("window", window, "cache", cache, "step %.3f ms (host %.3f)" % (dt / 10 * 1e3, th / 10 * 1e3), {k: round(v * 1e3, 3) for k, v in ph.items()})
        del p
