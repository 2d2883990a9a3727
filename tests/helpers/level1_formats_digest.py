"""Prints one sha256 per blend of a fixed set of deferred multi-band cycles (single and batched, the three precisions, CV_8UC3 and
CV_16SC3 tiles, 2 / 5 / 7 bands, a column window).  tests/test_gpu_level1_formats.py runs it twice - with the library's defaults and with
ISX_OUT12=0 ISX_G1P=0 (level 1 in 16-byte records, the layout before round 4's last change; ISX_G1Q8=0: planar level 1 without round 5's
Q8 records) - and compares the lines.  The `fmt` lines say which layout each single blend ran with."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import imagestitch_amd as G  # noqa: E402
from imagestitch_amd.blender import blend_batch  # noqa: E402

G.load()
rng = np.random.default_rng(2024)


def tiles_for(sizes, s16):
    out = []
    for (w, h) in sizes:
        if s16:
            img = rng.integers(-2000, 2001, (h, w, 3)).astype(np.int16)
        else:
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        mask = (rng.random((h, w)) > 0.2).astype(np.uint8) * 255
        mask[rng.random((h, w)) < 0.1] = 77
        out.append((torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()))
    return out


def digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a.cpu().numpy()).tobytes())
    return h.hexdigest()


RIGS = [([(-40, 7), (233, -12)], [(411, 300), (397, 290)]),
        ([(0, 0), (150, 9), (290, -6), (430, 4)], [(420, 260), (420, 255), (420, 262), (300, 250)]),
        ([(3, 1), (57, -2), (9, 77)], [(71, 93), (5, 140), (131, 33)])]
for prec in (0, 1, 2):
    for s16 in (False, True):
        for bands in (2, 5, 7):
            for ri, (corners, sizes) in enumerate(RIGS):
                tl = tiles_for(sizes, s16)
                mb = G.MultiBandBlender(False, bands, prec)
                mb.set_deferred_level0(True)
                mb.prepare(corners, sizes)
                for (ti, tm), c in zip(tl, corners):
                    mb.feed(ti, tm, c)
                d, m = mb.blend(out_f32=(prec != 0))
                lp = mb.last_path()
                print("single", prec, int(s16), bands, ri, lp["cycle"], lp["last_step"], digest(d, m))
                print("fmt", prec, int(s16), bands, ri, mb.level1_format())
    # the batched chain: three mosaics of one rig shape
    bl, ds, ms, keep = [], [], [], []
    for q in range(3):
        corners, sizes = [(0, 0), (200 + q, 5)], [(330, 210), (300, 220)]
        tl = tiles_for(sizes, False)
        keep.append(tl)
        mb = G.MultiBandBlender(False, 5, prec)
        mb.set_deferred_level0(True)
        mb.prepare(corners, sizes)
        for (ti, tm), c in zip(tl, corners):
            mb.feed(ti, tm, c)
        w, h = mb.result_size()
        bl.append(mb)
        ds.append(torch.empty((h, w, 3), dtype=torch.float32 if prec != 0 else torch.int16, device="cuda"))
        ms.append(torch.empty((h, w), dtype=torch.uint8, device="cuda"))
    blend_batch(bl, ds, ms)
    for q in range(3):
        print("batch", prec, q, digest(ds[q], ms[q]))
    # a column window of a pair
    corners, sizes = [(0, 0), (700, 11)], [(900, 400), (880, 390)]
    tl = tiles_for(sizes, False)
    mb = G.MultiBandBlender(False, 5, prec)
    mb.set_deferred_level0(True)
    mb.prepare(corners, sizes)
    mb.set_window(256, 1024)
    for (ti, tm), c in zip(tl, corners):
        mb.feed(ti, tm, c)
    d, m = mb.blend(out_f32=(prec != 0))
    print("window", prec, digest(d, m))
