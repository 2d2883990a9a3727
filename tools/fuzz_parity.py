#!/usr/bin/env python3
"""Randomised parity sweep: HIP path vs the CPU oracle on random geometries (run on the GPU box).
    python tools/fuzz_parity.py [seconds] [seed] [summary.json]
Every failure prints the seed of the case so that it can be replayed; exit code = number of failing cases.  With a third argument the
summary (cases, seconds, seed, mismatches per family) is also written as JSON - the file kept under profiles/.
tests/test_gpu_fuzz_slice.py runs run(20 s, fixed seed) under -m gpu."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import imagestitch_amd as G  # noqa: E402
from oracle import capi as O  # noqa: E402


def rot(rng, amp):
    a, b, c = rng.uniform(-amp, amp, 3)
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return (Ry @ Rx @ Rz).astype(np.float32)


def case_warp(rng):
    w, h = int(rng.integers(2, 400)), int(rng.integers(2, 300))
    f = float(rng.uniform(0.3, 3.0) * max(w, h))
    K = np.array([[f, 0, w / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), h / 2 + rng.uniform(-3, 3)], [0, 0, 1]], np.float32)
    R = rot(rng, 0.5)
    kind = int(rng.integers(0, 2))
    cn = int(rng.choice([1, 3]))
    src = rng.integers(0, 256, (h, w, 3) if cn == 3 else (h, w)).astype(np.uint8)
    interp, border = int(rng.integers(0, 2)), int(rng.choice([0, 1, 2, 4]))
    roi, _ = O.detect_roi(kind, f, K, R, w, h)
    if (roi[2] - roi[0] + 1) * (roi[3] - roi[1] + 1) > 4_000_000 or roi[2] < roi[0] or roi[3] < roi[1]:
        return "skip"
    wp = (G.CylindricalWarper() if kind == 0 else G.SphericalWarper()).create(f)
    c, d = wp.warp(src, K, R, interp, border)
    oc, od, _ = O.warp_u8(kind, f, K, R, src, interp, border)
    assert tuple(c) == tuple(oc), (c, oc)
    assert np.array_equal(d, od), np.argwhere(d != od)[:3]
    if cn == 3:
        c2, wi, wm = wp.warp_with_mask(src, K, R)
        _, oi, _ = O.warp_u8(kind, f, K, R, src, 1, 2)
        _, om, _ = O.warp_u8(kind, f, K, R, np.full((h, w), 255, np.uint8), 0, 0)
        assert np.array_equal(wi, oi) and np.array_equal(wm, om)


def case_blend(rng):
    n = int(rng.integers(1, 5))
    sizes = [(int(rng.integers(1, 180)), int(rng.integers(1, 150))) for _ in range(n)]
    corners = [(int(rng.integers(-60, 120)), int(rng.integers(-40, 90))) for _ in range(n)]
    bands, prec = int(rng.integers(0, 7)), int(rng.integers(0, 3))
    tiles = [(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), (rng.random((h, w)) > rng.uniform(0, 0.6)).astype(np.uint8) * 255) for (w, h) in sizes]
    deferred = bool(rng.integers(0, 2))
    u8 = bool(rng.integers(0, 2))
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0(deferred)
    ob = O.MultiBand(bands, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        if u8:
            mb.feed_u8(img, mask, c)
        else:
            mb.feed(img.astype(np.int16), mask, c)
        ob.feed(img.astype(np.int16), mask, c)
    of32 = prec != 0 and bool(rng.integers(0, 2))
    d, m = mb.blend(out_f32=of32)
    od, om = ob.blend(of32)
    assert np.array_equal(m, om)
    assert np.array_equal(d, od), (bands, prec, deferred, np.argwhere(d != od)[:3])


def case_feather(rng):
    n = int(rng.integers(1, 4))
    sizes = [(int(rng.integers(1, 200)), int(rng.integers(1, 160))) for _ in range(n)]
    corners = [(int(rng.integers(-60, 120)), int(rng.integers(-40, 90))) for _ in range(n)]
    sharp = float(rng.choice([0.02, 0.1, 0.5, 1.7]))
    fb, ob = G.FeatherBlender(False, sharp), O.Feather(sharp)
    fb.set_deferred_level0(bool(rng.integers(0, 2)))
    fb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    keep = []
    for (w, h), c in zip(sizes, corners):
        img = rng.integers(-300, 600, (h, w, 3)).astype(np.int16)
        mask = np.full((h, w), 255, np.uint8)
        mask[rng.random((h, w)) < rng.choice([0.0, 0.002, 0.3])] = 0
        keep.append((img, mask))
        fb.feed(img, mask, c)
        ob.feed(img, mask, c)
    d, m = fb.blend()
    od, om = ob.blend()
    assert np.array_equal(m, om) and np.array_equal(d, od)


def case_prep(rng):
    h, w = int(rng.integers(1, 300)), int(rng.integers(1, 400))
    kw, kh = int(rng.integers(1, 45)), int(rng.integers(1, 45))
    m = (rng.random((h, w)) < rng.choice([0.001, 0.05, 0.5])).astype(np.uint8) * int(rng.integers(1, 256))
    other = rng.integers(0, 256, (h, w)).astype(np.uint8)
    assert np.array_equal(G.dilate_and(m, kw, kh, other=other), O.dilate_rect(m, kw, kh) & other)
    img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    g = float(rng.choice([rng.uniform(0.2, 3.0), 1.0, 0.5, 2.0]))
    assert np.array_equal(G.gain_apply(img.copy(), g), O.gain_apply(img, g))


def case_seam(rng):
    from seam_cases import make_case
    try:
        c = _seam_case(rng, make_case)
    except AssertionError:
        return "skip"          # the generator rejects tiles that barely overlap
    ref, rh = O.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    got, gh = G.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    assert gh == rh and np.array_equal(got, ref)


def _seam_case(rng, make_case):
    return make_case(int(rng.integers(0, 1 << 30)), size1=(int(rng.integers(30, 200)), int(rng.integers(40, 260))), size2=(int(rng.integers(30, 200)), int(rng.integers(40, 260))),
                  tl1=(int(rng.integers(-40, 0)), int(rng.integers(-10, 10))), tl2=(int(rng.integers(0, 30)), int(rng.integers(-10, 10))), u8=bool(rng.integers(0, 2)),
                  horizontal=bool(rng.integers(0, 2)), swap=bool(rng.integers(0, 2)), holes=bool(rng.integers(0, 2)))


def case_blend_float_and_many(rng):
    """CV_32FC3 feeds in the float precisions, CV_8UC3 / CV_32FC3 outputs, up to 11 tiles (beyond one Cover / one TileSet)."""
    n = int(rng.integers(1, 12))
    sizes = [(int(rng.integers(2, 90)), int(rng.integers(2, 80))) for _ in range(n)]
    corners = [(int(rng.integers(-30, 200)), int(rng.integers(-30, 120))) for _ in range(n)]
    bands, prec = int(rng.integers(0, 6)), int(rng.integers(1, 3))
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0(bool(rng.integers(0, 2)))
    ob = O.MultiBand(bands, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    for (w, h), c in zip(sizes, corners):
        img = (rng.random((h, w, 3)) * 300 - 20).astype(np.float32)
        mask = (rng.random((h, w)) > 0.2).astype(np.uint8) * 255
        mb.feed(img, mask, c)
        ob.feed(img, mask, c)
    d, m = mb.blend(out_f32=True)
    od, om = ob.blend(True)
    assert np.array_equal(m, om) and np.array_equal(d, od), (n, bands, prec, np.argwhere(d != od)[:3])


def case_pipeline(rng):
    """PairStitcher: planned step (ROI verified on the device, scans scheduled inside blend) == host-synchronous step == oracle."""
    import torch
    from imagestitch_amd import synth
    from imagestitch_amd.pipeline import PairStitcher
    W, H = int(rng.integers(200, 900)), int(rng.integers(150, 600))
    F = float(rng.uniform(0.6, 1.5) * W)
    K, Rs = synth.camera_pair(W, H, F, yaw=float(rng.uniform(0.1, 0.35)))
    imgs = [rng.integers(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(2)]
    bands, prec = int(rng.integers(1, 6)), int(rng.integers(0, 3))
    t = [torch.from_numpy(a).cuda() for a in imgs]
    ps = PairStitcher(t, K, Rs, F, "cylindrical", bands, prec, 0, None, "int16", verify_at=int(rng.integers(-1, 4)))
    a, am = [x.clone() for x in ps.step()]
    a2, am2 = [x.clone() for x in ps.step()]
    b, bm = ps.step_sync()
    assert ps.check_plan() == 0
    assert torch.equal(a, b) and torch.equal(am, bm) and torch.equal(a2, b)
    corners, warped, seam = [], [], [s.cpu().numpy() for s in ps.seam]
    for i in range(2):
        c, wi, _ = O.warp_u8(O.CYL, F, K, Rs[i], imgs[i], 1, 2)
        corners.append(c); warped.append(wi)
    ob = O.MultiBand(bands, prec)
    ob.prepare(corners, [(w.shape[1], w.shape[0]) for w in warped])
    for i in range(2):
        ob.feed(warped[i].astype(np.int16), seam[i], corners[i])
    od, om = ob.blend(False)
    assert np.array_equal(b.cpu().numpy(), od) and np.array_equal(bm.cpu().numpy(), om)


def case_find(rng):
    """The whole DP seam finder: isx_dp_seam_find vs oracle/dpseam_np.py on 2-4 tiles with barrel-shaped, holed masks."""
    from oracle.dpseam_np import DpSeamFinder as OracleFinder
    from seam_cases import make_find_case
    n, u8 = int(rng.integers(2, 5)), bool(rng.integers(0, 2))
    images, corners, masks = make_find_case(int(rng.integers(0, 1 << 30)), n, u8, holes=bool(rng.integers(0, 2)),
                                            size=(int(rng.integers(20, 160)), int(rng.integers(30, 200))))
    ref = [m.copy() for m in masks]
    OracleFinder().find(images, corners, ref)
    got = [m.copy() for m in masks]
    G.DpSeamFinder().find(images, corners, got)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b), np.argwhere(a != b)[:3]


def case_warp_fused(rng):
    """The hot fused tile kernel (image LINEAR / REFLECT + mask NEAREST / CONSTANT of an all-255 mask, W:229 + W:232) on device
    tensors: dense and pitched destinations (per-pixel / dword stores), CV_8UC3 and CV_16SC3 outputs, cameras that look past the
    image (reflected taps on every side) and far past it (z <= 0 columns), both projectors."""
    import torch
    w, h = int(rng.integers(2, 700)), int(rng.integers(2, 500))
    f = float(rng.uniform(0.25, 3.0) * max(w, h))
    K = np.array([[f, 0, w / 2 + rng.uniform(-5, 5)], [0, f * rng.uniform(0.9, 1.1), h / 2 + rng.uniform(-5, 5)], [0, 0, 1]], np.float32)
    R = rot(rng, float(rng.choice([0.2, 0.6, 1.2])))
    kind = int(rng.integers(0, 2))
    src = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    roi, _ = O.detect_roi(kind, f, K, R, w, h)
    dw, dh = int(roi[2]) - int(roi[0]) + 1, int(roi[3]) - int(roi[1]) + 1
    if dw < 1 or dh < 1 or dw > 60000 or dh > 60000 or dw * dh > 4_000_000:
        return "skip"
    out16, pitched = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    wp = (G.CylindricalWarper() if kind == 0 else G.SphericalWarper()).create(f)
    t = torch.from_numpy(src).cuda()
    if pitched:
        es = 2 if out16 else 1
        pit = (dw * 3 * es + 63) // 64 * 64
        di = torch.zeros((dh * pit // es,), dtype=torch.int16 if out16 else torch.uint8, device="cuda").as_strided((dh, dw, 3), (pit // es, 3, 1))
        pm = (dw + 63) // 64 * 64
        dm = torch.zeros((dh * pm,), dtype=torch.uint8, device="cuda").as_strided((dh, dw), (pm, 1))
        c, wi, wm = wp.warp_with_mask(t, K, R, dst_img=di, dst_mask=dm)
    else:
        c, wi, wm = wp.warp_with_mask(t, K, R, out16=out16)
    oc, oi, _ = O.warp_u8(kind, f, K, R, src, 1, 2)
    _, om, _ = O.warp_u8(kind, f, K, R, np.full((h, w), 255, np.uint8), 0, 0)
    assert tuple(c) == tuple(oc)
    assert np.array_equal(wi.cpu().numpy(), oi.astype(np.int16) if out16 else oi), np.argwhere(wi.cpu().numpy() != oi)[:3]
    assert np.array_equal(wm.cpu().numpy(), om)


def case_linear_pair(rng):
    """A13, the reference's in-tree linear-ramp pair blend (B:141-717): random sizes (up to several LDS seam windows wide), offsets
    of either sign, dark regions so that every overlap class occurs."""
    import ctypes as C
    from imagestitch_amd import _lib
    big = rng.random() < 0.15
    h1, w1 = int(rng.integers(20, 900 if big else 200)), int(rng.integers(40, 1400 if big else 260))
    h2, w2 = h1 + int(rng.integers(-6, 7)), int(rng.integers(40, 1400 if big else 260))
    if h2 < 8:
        return "skip"
    img1 = rng.random((h1, w1, 3)).astype(np.float32) * 255
    img2 = rng.random((h2, w2, 3)).astype(np.float32) * 255
    for im in (img1, img2):                       # dark patches: the 1/0, 0/1 and 1/1 overlap classes (B:332-470)
        for _ in range(int(rng.integers(0, 4))):
            y, x = int(rng.integers(0, im.shape[0])), int(rng.integers(0, im.shape[1]))
            im[y:y + int(rng.integers(2, 40)), x:x + int(rng.integers(2, 60))] = float(rng.uniform(0, 8))
    ov = int(rng.integers(8, max(9, min(w1, w2) - 4)))
    tl1 = (int(rng.integers(-50, 50)), int(rng.integers(-50, 50)))
    tl2 = (tl1[0] + w1 - ov, tl1[1] + int(rng.integers(-6, 7)))
    rc, opano, oseam = O.blend_pair_linear(img1, img2, tl1, tl2)
    lib = _lib.load()
    pr, pc = C.c_int(), C.c_int()
    _lib.check(lib.isx_blend_pair_linear_size(h1, w1, h2, w2, tl1[0], tl1[1], tl2[0], tl2[1], C.byref(pr), C.byref(pc)))
    assert (pr.value, pc.value) == opano.shape[:2]
    pano = np.empty_like(opano)
    seam = np.zeros(opano.shape[0], np.int32)
    m1, m2, mp = _lib.as_mat(img1), _lib.as_mat(img2), _lib.as_mat(pano)
    grc = lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), tl1[0], tl1[1], tl2[0], tl2[1], C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None)
    assert (grc == 0) == (rc == 0), (grc, rc)
    if rc != 0:
        return "skip"                            # no overlap: both sides return without a panorama (B:182-183)
    assert np.array_equal(seam, oseam)
    assert np.array_equal(pano, opano, equal_nan=True), np.argwhere(pano != opano)[:5]


def case_strip(rng):
    """One column strip of a row of tiles (isx_blender_set_window) against the ORACLE's whole blend: random tiles, bands, precision,
    input type and window; only the tiles mosaic.tiles_for_window lists are fed to the HIP blender."""
    from imagestitch_amd import mosaic
    n = int(rng.integers(1, 7))
    sizes = [(int(rng.integers(40, 260)), int(rng.integers(8, 90))) for _ in range(n)]
    x, corners = int(rng.integers(-50, 50)), []
    for w, _ in sizes:
        corners.append((x, int(rng.integers(-12, 12))))
        x += int(rng.integers(max(w // 4, 1), w + 20))
    bands, prec = int(rng.integers(1, 6)), int(rng.integers(0, 3))
    as_f32 = prec != 0 and bool(rng.integers(0, 2))
    imgs, masks = [], []
    for w, h in sizes:
        imgs.append((rng.random((h, w, 3)) * 300 - 20).astype(np.float32) if as_f32 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        m = (rng.random((h, w)) > 0.15).astype(np.uint8) * 255
        masks.append(m)
    ob = O.MultiBand(bands, prec)
    ob.prepare(corners, sizes)
    for im, m, c in zip(imgs, masks, corners):
        ob.feed(im if as_f32 else im.astype(np.int16), m, c)
    out_f32 = prec != 0 and bool(rng.integers(0, 2))
    od, om = ob.blend(out_f32)
    fw = od.shape[1]
    x0 = int(rng.integers(0, (fw - 1) // 128 + 1)) * 128
    x1 = x0 + int(rng.integers(1, 4)) * 128 if rng.integers(0, 2) else x0 + int(rng.integers(1, 400))
    act = mosaic.tiles_for_window(corners, sizes, bands, x0, x1)
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0(True)
    mb.set_window(x0, x1)
    mb.prepare(corners, sizes)
    if not act:
        return "skip"
    as_u8 = bool(rng.integers(0, 2))      # one input type per cycle: a tile of another type ends the deferred cycle, and with it the window
    for i in act:
        if as_f32:
            mb.feed(imgs[i], masks[i], corners[i])
        elif as_u8:
            mb.feed_u8(imgs[i], masks[i], corners[i])
        else:
            mb.feed(imgs[i].astype(np.int16), masks[i], corners[i])
    d, m = mb.blend(out_f32=out_f32)
    xe = min(x1, fw)
    assert d.shape[1] == x1 - x0
    assert np.array_equal(m[:, :xe - x0], om[:, x0:xe]) and np.array_equal(d[:, :xe - x0], od[:, x0:xe]), (n, bands, prec, x0, x1, act)


def case_batch(rng):
    """isx_blender_blend_batch: 2-7 blenders of one rig family (same bands / precision / input type, different tiles, sizes and corners;
    now and then one that cannot share the chain: eager, windowed, or other bands) blended in ONE call - every result against the ORACLE's."""
    from imagestitch_amd.blender import blend_batch
    nb = int(rng.integers(2, 8))
    bands, prec = int(rng.integers(1, 6)), int(rng.integers(0, 3))
    as_u8 = bool(rng.integers(0, 2))
    out_f32 = prec != 0 and bool(rng.integers(0, 2))
    blenders, expect, dsts, dmasks = [], [], [], []
    for b in range(nb):
        odd = int(rng.integers(0, 8)) == 0                     # a blender that does not qualify for the shared chain
        n = int(rng.integers(1, 4))
        sizes = [(int(rng.integers(40, 220)), int(rng.integers(8, 90))) for _ in range(n)]
        x, corners = int(rng.integers(-50, 50)), []
        for w, _ in sizes:
            corners.append((x, int(rng.integers(-12, 12))))
            x += int(rng.integers(max(w // 4, 1), w + 20))
        bb = bands if not (odd and rng.integers(0, 2)) else int(rng.integers(1, 6))
        imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for w, h in sizes]
        masks = [(rng.random((h, w)) > 0.15).astype(np.uint8) * 255 for w, h in sizes]
        ob = O.MultiBand(bb, prec)
        ob.prepare(corners, sizes)
        for im, m, c in zip(imgs, masks, corners):
            ob.feed(im.astype(np.int16), m, c)
        expect.append(ob.blend(out_f32))
        mb = G.MultiBandBlender(False, bb, prec)
        mb.set_deferred_level0(not (odd and bb == bands))      # the odd one with the family's bands runs the eager cycle
        mb.prepare(corners, sizes)
        for im, m, c in zip(imgs, masks, corners):
            if as_u8:
                mb.feed_u8(im, m, c)
            else:
                mb.feed(im.astype(np.int16), m, c)
        blenders.append(mb)
        w, h = mb.result_size()
        dsts.append(np.full((h, w, 3), -5, np.float32 if out_f32 else np.int16))
        dmasks.append(np.full((h, w), 7, np.uint8))
    blend_batch(blenders, dsts, dmasks)
    for b in range(nb):
        assert np.array_equal(dmasks[b], expect[b][1]) and np.array_equal(dsts[b], expect[b][0]), (nb, b, bands, prec, as_u8, out_f32)


def case_strip_feather(rng):
    """One column strip of a FeatherBlender mosaic against the oracle's whole blend; only the tiles that overlap the strip are fed."""
    n = int(rng.integers(1, 7))
    sizes = [(int(rng.integers(30, 260)), int(rng.integers(8, 90))) for _ in range(n)]
    x, corners = int(rng.integers(-50, 50)), []
    for w, _ in sizes:
        corners.append((x, int(rng.integers(-12, 12))))
        x += int(rng.integers(max(w // 4, 1), w + 20))
    sharp = float(rng.choice([0.02, 0.1, 0.5]))
    u8 = bool(rng.integers(0, 2))
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if u8 else rng.integers(-300, 600, (h, w, 3)).astype(np.int16) for w, h in sizes]
    masks = []
    for w, h in sizes:
        m = np.full((h, w), 255, np.uint8)
        m[rng.random((h, w)) < rng.choice([0.0, 0.002, 0.3])] = 0
        masks.append(m)
    ob = O.Feather(sharp)
    ob.prepare(corners, sizes)
    for im, m, c in zip(imgs, masks, corners):
        ob.feed(im.astype(np.int16), m, c)
    od, om = ob.blend()
    fw = od.shape[1]
    x0 = int(rng.integers(0, (fw - 1) // 128 + 1)) * 128
    x1 = x0 + int(rng.integers(1, 400))
    rx = min(c[0] for c in corners)
    act = [i for i in range(n) if corners[i][0] - rx < x1 and corners[i][0] - rx + sizes[i][0] > x0]
    if not act:
        return "skip"
    fb = G.FeatherBlender(False, sharp)
    fb.set_deferred_level0(True)
    fb.set_window(x0, x1)
    fb.prepare(corners, sizes)
    for i in act:
        (fb.feed_u8 if u8 else fb.feed)(imgs[i], masks[i], corners[i])
    d, m = fb.blend()
    xe = min(x1, fw)
    assert d.shape[1] == x1 - x0
    assert np.array_equal(m[:, :xe - x0], om[:, x0:xe]) and np.array_equal(d[:, :xe - x0], od[:, x0:xe]), (n, x0, x1, act)


def case_s16_tiles(rng):
    """Round 4: CV_16SC3 tiles of the whole int16 range through the deferred cycle (k_pyr_down0 / k_collapse_roll take them), device mats at
    odd alignments, masks that are not just 0 / 255, two to five tiles over one place, every precision and result type."""
    import torch
    n = int(rng.integers(1, 6))
    big = bool(rng.integers(0, 2))
    sizes = [(int(rng.integers(2, 420 if big else 120)), int(rng.integers(2, 300 if big else 100))) for _ in range(n)]
    spread = 260 if big else 70
    corners = [(int(rng.integers(-spread, spread)), int(rng.integers(-40, 60))) for _ in range(n)]
    bands, prec = int(rng.integers(1, 7)), int(rng.integers(0, 3))
    lo, hi = [(-32768, 32767), (0, 255), (-2000, 3000)][int(rng.integers(0, 3))]
    mode = [True, "copy", False][int(rng.integers(0, 3))]
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0(mode)
    ob = O.MultiBand(bands, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    keep = []
    dev = bool(rng.integers(0, 2))
    for (w, h), c in zip(sizes, corners):
        img = rng.integers(lo, hi + 1, (h, w, 3)).astype(np.int16)
        mask = rng.integers(0, 256, (h, w)).astype(np.uint8)
        mask[rng.random((h, w)) < rng.uniform(0, 0.7)] = 0
        mask[rng.random((h, w)) < rng.uniform(0, 0.7)] = 255
        ob.feed(img, mask, c)
        if dev:
            pad, off = int(rng.integers(0, 9)), int(rng.integers(0, 4))
            pitch = w * 3 + pad
            buf = torch.zeros((h * pitch + off + 8,), dtype=torch.int16, device="cuda")
            ti = buf[off:].as_strided((h, w, 3), (pitch, 3, 1))
            ti.copy_(torch.from_numpy(img).cuda())
            tm = torch.from_numpy(mask).cuda()
            keep.append((buf, ti, tm))
            mb.feed(ti, tm, c)
        else:
            mb.feed(img, mask, c)
    kind = int(rng.integers(0, 3)) if prec != 0 else int(rng.integers(0, 2)) * 2
    d, m = mb.blend(out_f32=kind == 1, out_u8=kind == 2)
    d = d.cpu().numpy() if hasattr(d, "cpu") else d
    m = m.cpu().numpy() if hasattr(m, "cpu") else m
    od, om = ob.blend(kind == 1)
    if kind == 2:
        od = np.clip(od, 0, 255).astype(np.uint8)
    assert np.array_equal(m, om)
    assert np.array_equal(d, od), (bands, prec, mode, n, np.argwhere(d != od)[:3])


def case_round4_calls(rng):
    """Round 4's new entries: Blender::NO, isx_convert_to, the gain folded into the fused warp, the mask preparation folded into the feed, and
    the two warps of a tile as calls of their own (image-only / mask-only tile kernels) - each against the oracle's separate stages."""
    which = int(rng.integers(0, 4))
    if which == 0:      # Blender::NO
        n = int(rng.integers(1, 5))
        sizes = [(int(rng.integers(1, 150)), int(rng.integers(1, 120))) for _ in range(n)]
        corners = [(int(rng.integers(-50, 90)), int(rng.integers(-40, 60))) for _ in range(n)]
        nb, ob = G.NoBlender(), O.NoBlend()
        nb.prepare(corners, sizes); ob.prepare(corners, sizes)
        for (w, h), c in zip(sizes, corners):
            img = rng.integers(-32768, 32768, (h, w, 3)).astype(np.int16)
            mask = rng.integers(0, 256, (h, w)).astype(np.uint8) * (rng.random((h, w)) < 0.6)
            nb.feed(img, mask.astype(np.uint8), c); ob.feed(img, mask.astype(np.uint8), c)
        d, m = nb.blend(); od, om = ob.blend()
        assert np.array_equal(m, om) and np.array_equal(d, od)
    elif which == 1:    # convertTo
        h, w, cn = int(rng.integers(1, 90)), int(rng.integers(1, 130)), int(rng.choice([1, 3]))
        f = (rng.standard_normal((h, w, cn) if cn == 3 else (h, w)) * float(rng.choice([3.0, 300.0, 40000.0, 1e12]))).astype(np.float32)
        f.flat[::17] = np.round(f.flat[::17]) + 0.5
        if cn == 3:
            assert np.array_equal(G.convert_to(f, np.int16), O.convert_f32(f, np.int16))
            u = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            assert np.array_equal(G.convert_to(u, np.int16), u.astype(np.int16)) and np.array_equal(G.convert_to(u, np.float32), u.astype(np.float32))
            s16 = rng.integers(-500, 900, (h, w, 3)).astype(np.int16)
            assert np.array_equal(G.convert_to(s16, np.uint8), np.clip(s16, 0, 255).astype(np.uint8))
        else:
            u = rng.integers(0, 256, (h, w)).astype(np.uint8)
            assert np.array_equal(G.convert_to(u, np.float32), u.astype(np.float32))
    elif which == 2:    # gain in the warp + dilate in the feed, through a blend
        w, h = int(rng.integers(8, 300)), int(rng.integers(8, 220))
        f = float(rng.uniform(0.5, 2.0) * max(w, h))
        K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
        kind = int(rng.integers(0, 2))
        Rs = [rot(rng, 0.15), rot(rng, 0.15)]
        wp = (G.CylindricalWarper() if kind == 0 else G.SphericalWarper()).create(f)
        gains = [float(rng.uniform(0.6, 1.6)), 1.0]
        kw, kh = int(rng.integers(1, 34)), int(rng.integers(1, 34))
        typ = int(rng.integers(0, 3))
        nbands = int(rng.integers(1, 5))
        b = [G.MultiBandBlender(False, nbands, 0), G.FeatherBlender(False, 0.1), G.NoBlender()][typ]
        if typ < 2:
            b.set_deferred_level0([False, True, "copy"][int(rng.integers(0, 3))])
        corners, sizes, parts = [], [], []
        for i in range(2):
            src = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
            roi, _ = O.detect_roi(kind, f, K, Rs[i], w, h)
            if (roi[2] - roi[0] + 1) * (roi[3] - roi[1] + 1) > 2_000_000 or roi[2] < roi[0] or roi[3] < roi[1]:
                return "skip"
            wp.set_gain(gains[i])
            c, wi, wm = wp.warp_with_mask(src, K, Rs[i])
            oc, owi, _ = O.warp_u8(kind, f, K, Rs[i], src, 1, 2)
            _, owm, _ = O.warp_u8(kind, f, K, Rs[i], np.full((h, w), 255, np.uint8), 0, 0)
            owi = O.gain_apply(owi, gains[i])
            assert tuple(c) == tuple(oc) and np.array_equal(wi, owi) and np.array_equal(wm, owm)
            seam = (rng.random(wm.shape) < rng.choice([0.001, 0.02, 0.4])).astype(np.uint8) * 255
            corners.append(tuple(c)); sizes.append((wm.shape[1], wm.shape[0])); parts.append((wi, seam, wm))
        ob = [O.MultiBand(nbands, 0), O.Feather(0.1), O.NoBlend()][typ]
        b.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        for (wi, seam, wm), c in zip(parts, corners):
            b.feed_dilated(wi.astype(np.int16), seam, wm, kw, kh, c)
            ob.feed(wi.astype(np.int16), O.dilate_rect(seam, kw, kh) & wm, c)
        d, m = b.blend()
        od, om = ob.blend(False) if typ == 0 else ob.blend()
        assert np.array_equal(m, om) and np.array_equal(d, od), (typ, kw, kh)
    else:               # the two warps of a tile as calls of their own, device mats with odd pitches
        import torch
        w, h = int(rng.integers(2, 500)), int(rng.integers(3, 360))
        f = float(rng.uniform(0.3, 3.0) * max(w, h))
        K = np.array([[f, 0, w / 2 + rng.uniform(-3, 3)], [0, f * rng.uniform(0.9, 1.1), h / 2 + rng.uniform(-3, 3)], [0, 0, 1]], np.float32)
        R = rot(rng, 0.6)
        kind = int(rng.integers(0, 2))
        roi, _ = O.detect_roi(kind, f, K, R, w, h)
        if (roi[2] - roi[0] + 1) * (roi[3] - roi[1] + 1) > 3_000_000 or roi[2] < roi[0] or roi[3] < roi[1]:
            return "skip"
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        mask = rng.integers(0, 256, (h, w)).astype(np.uint8)
        wp = (G.CylindricalWarper() if kind == 0 else G.SphericalWarper()).create(f)
        _, owi, _ = O.warp_u8(kind, f, K, R, img, 1, 2)
        _, owm, _ = O.warp_u8(kind, f, K, R, mask, 0, 0)
        dh, dw = owm.shape
        pi, pm = dw * 3 + int(rng.integers(0, 7)), dw + int(rng.integers(0, 7))
        di = torch.zeros((dh * pi + 5,), dtype=torch.uint8, device="cuda")[int(rng.integers(0, 4)):][:dh * pi].as_strided((dh, dw, 3), (pi, 3, 1))
        dm = torch.zeros((dh * pm + 5,), dtype=torch.uint8, device="cuda")[int(rng.integers(0, 4)):][:dh * pm].as_strided((dh, dw), (pm, 1))
        wp.warp(torch.from_numpy(img).cuda(), K, R, 1, 2, dst=di)
        wp.warp(torch.from_numpy(mask).cuda(), K, R, 0, 0, dst=dm)
        assert np.array_equal(di.cpu().numpy(), owi) and np.array_equal(dm.cpu().numpy(), owm)




def case_many_tiles(rng):
    """Round 4: more than 20 tiles in one deferred multi-band cycle - blend() cuts the result into column strips that at most 20 tiles reach
    (run_blend_deferred_strips), or falls back to the eager cycle when a strip is reached by more (tiles stacked); rows with random overlaps,
    sometimes two rows, both tile types, every precision, references and private copies, sometimes inside a caller's column window."""
    import torch
    n = int(rng.integers(21, 45))
    s16 = bool(rng.integers(0, 2))
    rows = int(rng.integers(1, 3))
    corners, sizes, x = [], [], 0
    for i in range(n):
        w, h = int(rng.integers(8, 90)), int(rng.integers(6, 70))
        corners.append((x, int(rng.integers(-10, 11)) + (i % rows) * int(rng.integers(20, 60))))
        sizes.append((w, h))
        x += int(w * rng.uniform(0.0, 0.9))         # 0: a tile over its neighbour (many over one place)
    bands, prec = int(rng.integers(1, 6)), int(rng.integers(0, 3))
    mode = [True, "copy"][int(rng.integers(0, 2))]
    # round 5: one chain over a device-resident table of the tiles (default); one case in three runs round 4's column strips (ISX_TAB=0, read per blend)
    os.environ["ISX_TAB"] = "0" if rng.integers(0, 3) == 0 else "1"
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0(mode)
    ob = O.MultiBand(bands, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    fw, _ = mb.result_size()
    win = None
    if fw > 400 and rng.integers(0, 3) == 0:
        x0 = int(rng.integers(0, fw // 256)) * 128
        x1 = min(x0 + int(rng.integers(1, 6)) * 128, (fw // 128) * 128)
        if x1 > x0:
            win = (x0, x1)
            mb.set_window(x0, x1)
    keep = []
    for (w, h), c in zip(sizes, corners):
        img = rng.integers(-3000, 3001, (h, w, 3)).astype(np.int16) if s16 else rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        mask = rng.integers(0, 256, (h, w)).astype(np.uint8)
        mask[rng.random((h, w)) < rng.uniform(0, 0.5)] = 0
        mask[rng.random((h, w)) < rng.uniform(0, 0.7)] = 255
        ob.feed(img.astype(np.int16), mask, c)
        ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
        keep.append((ti, tm))
        if s16:
            mb.feed(ti, tm, c)
        else:
            mb.feed_u8(ti, tm, c)
        if mode == "copy":
            ti.fill_(1), tm.fill_(2)
    f32 = prec != 0 and bool(rng.integers(0, 2))
    try:
        d, m = mb.blend(out_f32=f32)
    except Exception as e:      # the one refusal of the path: a caller's window with more than 20 tiles over one 128-column strip (no eager cycle to fall back to)
        if win and "more than 20 tiles reach" in str(e):
            return "skip"
        raise
    finally:
        os.environ.pop("ISX_TAB", None)
    d, m = d.cpu().numpy(), m.cpu().numpy()
    od, om = ob.blend(f32)
    if win:
        od, om = od[:, win[0]:win[1]], om[:, win[0]:win[1]]
    assert np.array_equal(m, om)
    assert np.array_equal(d, od), (n, bands, prec, mode, win, mb.last_path(), np.argwhere(d != od)[:3])


def case_fused_feed(rng):
    """Round 5: feed() of mode 2 as ONE pass (k_feed_strip / k_feed_pd0: level 1 + the private copy out of one read of the caller's tile) with the
    copy of a CV_16SC3 tile narrowed to CV_8UC3 while its values are bytes.  Two to four CYCLES of one blender whose tiles are all bytes, bytes but
    for one value somewhere (an escaped segment: the cycle is widened before the last step), or full-range shorts (the blender keeps wide copies
    from the first violation on), CV_16SC3 or CV_8UC3 tiles, random sizes and offsets (rim strips everywhere), 1-6 bands, every precision, the
    caller's mats poisoned after feed(), now and then a column window or level introspection between feed and blend."""
    import torch
    n = int(rng.integers(1, 5))
    big = bool(rng.integers(0, 3))
    sizes = [(int(rng.integers(2, 700 if big else 90)), int(rng.integers(2, 400 if big else 80))) for _ in range(n)]
    spread = 420 if big else 60
    corners = [(int(rng.integers(-spread, spread)), int(rng.integers(-50, 70))) for _ in range(n)]
    bands, prec = int(rng.integers(1, 7)), int(rng.integers(0, 3))
    u8 = rng.integers(0, 4) == 0
    mb = G.MultiBandBlender(False, bands, prec)
    mb.set_deferred_level0("copy")
    violated = False
    for cycle in range(int(rng.integers(2, 5))):
        kind = "bytes" if u8 else ["bytes", "bytes", "one", "full"][int(rng.integers(0, 4))]
        ob = O.MultiBand(bands, prec)
        mb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        fw, _ = mb.result_size()
        win = None
        if fw >= 256 and rng.integers(0, 4) == 0:
            x0 = int(rng.integers(0, fw // 128)) * 128
            x1 = min(x0 + int(rng.integers(1, 4)) * 128, fw + 64)
            win = (x0, x1)
        any_bad = False
        for (w, h), c in zip(sizes, corners):
            if kind == "full":
                img = rng.integers(-32768, 32768, (h, w, 3)).astype(np.int16)
            else:
                img = rng.integers(0, 256, (h, w, 3)).astype(np.int16)
                if kind == "one" and rng.integers(0, 2):
                    img[int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 3))] = [256, -1, 32767, -32768, 300][int(rng.integers(0, 5))]
            any_bad = any_bad or bool(((img < 0) | (img > 255)).any())
            mask = rng.integers(0, 256, (h, w)).astype(np.uint8)
            mask[rng.random((h, w)) < rng.uniform(0, 0.7)] = 0
            mask[rng.random((h, w)) < rng.uniform(0, 0.7)] = 255
            ob.feed(img, mask, c)
            if u8:
                ti = torch.from_numpy(img.astype(np.uint8)).cuda()
            else:       # a device view at an odd alignment (a 2-byte aligned row start is all the windows may assume)
                pad, off = int(rng.integers(0, 9)), int(rng.integers(0, 4))
                pitch = w * 3 + pad
                buf = torch.zeros((h * pitch + off + 8,), dtype=torch.int16, device="cuda")
                ti = buf[off:].as_strided((h, w, 3), (pitch, 3, 1))
                ti.copy_(torch.from_numpy(img).cuda())
            tm = torch.from_numpy(mask).cuda()
            (mb.feed_u8 if u8 else mb.feed)(ti, tm, c)
            ti.fill_(77 if u8 else -7), tm.fill_(99)
        if win is None and bands <= 4 and rng.integers(0, 6) == 0:
            lvl = int(rng.integers(0, mb.numBands() + 1))
            gl, gw = mb.level(lvl)
            ol, ow = ob.level(lvl)
            assert np.array_equal(gl, ol) and np.array_equal(gw, ow), ("level", lvl)
        if win:
            mb.set_window(*win)
        f32 = prec != 0 and bool(rng.integers(0, 2))
        d, m = mb.blend(out_f32=f32)
        if win:
            mb.set_window(0, 0)
        d, m = d.cpu().numpy(), m.cpu().numpy()
        od, om = ob.blend(f32)
        if win:
            od, om = od[:, win[0]:win[1]], om[:, win[0]:win[1]]
            d, m = d[:, :od.shape[1]], m[:, :om.shape[1]]
        path = mb.feed_path()
        assert np.array_equal(m, om), (cycle, kind)
        assert np.array_equal(d, od), (cycle, kind, n, bands, prec, path, np.argwhere(d != od)[:3])
        if not u8 and path["fused_tiles"] == n:
            want = "none" if violated else ("widened" if any_bad else "confirmed")
            assert path["narrowed"] == want, (path, want, cycle, kind)
            violated = violated or any_bad


def case_round6_calls(rng):
    """Round 6: (a) 1 - 9 tiles' fused warps collected by isx_warper_begin_batch and launched as one kernel per variant (blockIdx.z = tile; random
    sizes, cameras, CV_8UC3 / CV_16SC3 outputs, dense / pitched rows, now and then a call in the middle that cannot be collected) - every tile
    against the oracle's warp; (b) isx_warper_roi, which ranks the border on the caller's thread where the extrema provably lie there, against the
    oracle's scan of every source pixel (ROI and float extrema), the host-only self-test entry included."""
    import ctypes as C
    import torch
    kind = int(rng.integers(0, 2))
    f = float(rng.uniform(150.0, 900.0))
    wp = (G.CylindricalWarper() if kind == 0 else G.SphericalWarper()).create(f)
    if rng.integers(0, 2):
        wp.set_deferred_verify(True)
    nt = int(rng.integers(1, 10))
    jobs = []
    for i in range(nt):
        w, h = int(rng.integers(2, 420)), int(rng.integers(2, 300))
        K = np.array([[f * rng.uniform(0.8, 1.2), 0, w / 2 + rng.uniform(-5, 5)], [0, f * rng.uniform(0.8, 1.2), h / 2 + rng.uniform(-5, 5)], [0, 0, 1]], np.float32)
        R = rot(rng, float(rng.choice([0.2, 0.6, 1.2])))
        oroi, omm = O.detect_roi(kind, f, K, R, w, h)
        dw, dh = int(oroi[2]) - int(oroi[0]) + 1, int(oroi[3]) - int(oroi[1]) + 1
        if dw < 1 or dh < 1 or dw > 60000 or dh > 60000 or dw * dh > 1_500_000:
            continue
        roi, mm = wp.warpRoi((w, h), K, R, with_minmax=True)
        assert tuple(int(v) for v in roi) == tuple(int(v) for v in oroi), (roi, oroi)
        assert np.array_equal(np.asarray(mm, np.float32), omm), (mm, omm)
        # the host-only entry: the same answer wherever it answers at all
        hr, hm = np.zeros(4, np.int32), np.zeros(4, np.float32)
        fp = C.POINTER(C.c_float)
        Kc, Rc = np.ascontiguousarray(K.reshape(9)), np.ascontiguousarray(R.reshape(9))
        rc = G.load().isx_selftest_roi_host(kind, C.c_float(f), Kc.ctypes.data_as(fp), Rc.ctypes.data_as(fp), w, h, int(rng.integers(0, 2)),
                                            hr.ctypes.data_as(C.POINTER(C.c_int)), hm.ctypes.data_as(fp))
        if rc == 0:
            assert np.array_equal(hr, oroi) and np.array_equal(hm, omm), (hr, oroi, hm, omm)
        src = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        out16, pitched = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        es = 2 if out16 else 1
        pit = (dw * 3 * es + 63) // 64 * 64 if pitched else dw * 3 * es
        di = torch.full((dh * pit // es,), 77, dtype=torch.int16 if out16 else torch.uint8, device="cuda").as_strided((dh, dw, 3), (pit // es, 3, 1))
        pm = (dw + 63) // 64 * 64 if pitched else dw
        dm = torch.full((dh * pm,), 99, dtype=torch.uint8, device="cuda").as_strided((dh, dw), (pm, 1))
        smask = None
        if rng.random() < 0.15:          # a caller-supplied source mask: launched at once, not collected
            smask = (rng.integers(0, 4, (h, w)) > 0).astype(np.uint8) * 255
        jobs.append((torch.from_numpy(src).cuda(), K, R, [int(v) for v in oroi], di, dm, src, out16, smask, (w, h)))
    if not jobs:
        return "skip"
    wp.begin_batch()
    for (t, K, R, roi, di, dm, src, out16, smask, _) in jobs:
        wp.warp_with_mask_planned(t, K, R, roi, di, dm, mask=None if smask is None else torch.from_numpy(smask).cuda())
    wp.end_batch()
    wp.plan_status()
    for (t, K, R, roi, di, dm, src, out16, smask, (w, h)) in jobs:
        _, oi, _ = O.warp_u8(kind, f, K, R, src, 1, 2)
        _, om, _ = O.warp_u8(kind, f, K, R, np.full((h, w), 255, np.uint8) if smask is None else smask, 0, 0)
        assert np.array_equal(di.cpu().numpy(), oi.astype(np.int16) if out16 else oi), np.argwhere(di.cpu().numpy() != oi)[:3]
        assert np.array_equal(dm.cpu().numpy(), om)


CASES = [case_warp, case_blend, case_feather, case_prep, case_seam, case_blend_float_and_many, case_pipeline, case_find, case_warp_fused,
         case_linear_pair, case_strip, case_strip_feather, case_batch, case_s16_tiles, case_round4_calls, case_many_tiles, case_fused_feed, case_round6_calls]


def run(budget, seed0, verbose=True, progress_path=None):
    """Round-robin over the case families for `budget` seconds; case n uses seed seed0 * 1000003 + n.  Returns the summary dict.
    progress_path: the summary so far is written there every two minutes ("partial": true), so that a soak the GPU box's time limit cuts
    short still leaves its count behind (round 6 lost an hour-long one that way)."""
    G.load()
    t0, n, bad, skipped = time.time(), 0, 0, 0
    only = os.environ.get("ISX_FUZZ_ONLY", "")       # a comma-separated subset of the families (a soak of what a round changed)
    CASES = [f for f in globals()["CASES"] if not only or f.__name__ in only.split(",")]
    counts = {f.__name__: 0 for f in CASES}
    fails = {f.__name__: 0 for f in CASES}
    failing_seeds = []
    def summary(partial):
        d = {"cases": n, "seconds": round(time.time() - t0, 1), "seed": seed0, "mismatches": bad, "skipped_geometries": int(skipped),
             "per_family": {k: {"cases": counts[k], "mismatches": fails[k]} for k in counts}, "failing_seeds": failing_seeds[:50]}
        if partial:
            d["partial"] = True
        return d
    last_dump = t0
    while time.time() - t0 < budget:
        if progress_path and time.time() - last_dump > 120.0:
            import json
            with open(progress_path, "w") as f:
                json.dump(summary(True), f, indent=1)
            last_dump = time.time()
        fn = CASES[n % len(CASES)]
        seed = seed0 * 1000003 + n
        try:
            r = fn(np.random.default_rng(seed))
            skipped += r == "skip"
            counts[fn.__name__] += 1
        except AssertionError:
            bad += 1; fails[fn.__name__] += 1; failing_seeds.append([fn.__name__, seed])
            if verbose:
                print("FAIL", fn.__name__, "seed", seed)
                traceback.print_exc(limit=2)
        except Exception as e:   # geometry the reference itself rejects must be rejected the same way on both sides
            try:
                code = getattr(e, "code", None)
            except Exception:
                code = None
            if verbose:
                print("EXC ", fn.__name__, "seed", seed, type(e).__name__, code, str(e)[:120])
            bad += 1; fails[fn.__name__] += 1; failing_seeds.append([fn.__name__, seed])
        n += 1
    return summary(False)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = run(budget, seed0, progress_path=sys.argv[3] if len(sys.argv) > 3 else None)
    print("cases", out["cases"], {k: v["cases"] for k, v in out["per_family"].items()}, "skipped", out["skipped_geometries"],
          "failures", out["mismatches"], "in %.0f s" % out["seconds"])
    if len(sys.argv) > 3:
        import json
        import subprocess
        try:
            out["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
        except Exception:
            pass
        try:
            import torch
            out["device"] = torch.cuda.get_device_name(0)
        except Exception:
            pass
        with open(sys.argv[3], "w") as f:
            json.dump(out, f, indent=1)
    return out["mismatches"]


if __name__ == "__main__":
    sys.exit(min(main(), 100))
