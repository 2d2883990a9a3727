"""The one artefact of the reference whose inputs can be reconstructed.

`images_warped_f[0].bmp` (written at S:1195 after warp S:1156 and gain compensation S:1165-1171) is
warp(`src2.bmp`) with f = 2707.47 (the scale hard-coded at W:30), principal point 550.5, R = I, multiplied by the
GainCompensator's gain (0.98872).  A crop of it and the source window it samples are committed as
tests/golden/ref_warp_artifact.npz (tests/golden/make_golden.py).

What the comparison establishes:
  * corner, ROI and size of the warped tile, the back-projection maps, the 1/32-pixel coordinate quantisation, the
    bilinear weights, the border handling and the gain step (saturate_cast<uchar>(round-half-even(v * gain))) of the
    oracle reproduce the reference's own output;
  * the author's binary ran OpenCV's OpenCL (T-API / UMat) remap: float blend of the four taps, rounded half-to-EVEN.
    With that rounding 99.98 % of the values are identical (the rest are +-1 at pixels whose map coordinate sits on a
    1/64-pixel quantisation boundary: device sin / cos ulps);
  * OpenCV's CPU remap — the spec of record of this build — is the same sum rounded half-UP
    ((sum + 2^14) >> 15): it differs from the artefact exactly where the weighted sum is a tie.
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def art():
    return np.load(os.path.join(HERE, "golden", "ref_warp_artifact.npz"))


def _maps(oracle, art):
    f = float(art["focal"])
    c = float(art["centre"])
    K = np.array([[f, 0, c], [0, f, c], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    w, h = [int(v) for v in art["src_size"]]
    roi, _ = oracle.detect_roi(oracle.CYL, f, K, R, w, h)
    assert np.array_equal(roi, art["roi"])
    aw, ah = [int(v) for v in art["artifact_size"]]
    assert (roi[2] - roi[0] + 1, roi[3] - roi[1] + 1) == (aw, ah)          # the committed bitmap IS dst.create(roi.h + 1, roi.w + 1)
    ox, oy = [int(v) for v in art["crop_origin"]]
    crop = art["artifact_crop"]
    _, _, _, k_rinv = oracle.camera(K, R)
    sub = np.array([roi[0] + ox, roi[1] + oy, roi[0] + ox + crop.shape[1] - 1, roi[1] + oy + crop.shape[0] - 1], np.int32)
    xm, ym = oracle.build_maps(oracle.CYL, f, k_rinv, sub)
    sx0, sy0 = [int(v) for v in art["src_origin"]]
    return xm - np.float32(sx0), ym - np.float32(sy0)      # exact: the window origin is a small integer


def _tie_even_bilinear(src, xm, ym):
    """OpenCV's OpenCL remap (remap.cl, INTER_LINEAR): weights rint(frac * 32) / 32, float blend, convert_uchar_sat_rte."""
    x0 = np.floor(xm).astype(np.int64)
    y0 = np.floor(ym).astype(np.int64)
    ux = (np.rint((xm - np.floor(xm)) * np.float32(32)) / 32.0).astype(np.float64)[..., None]
    uy = (np.rint((ym - np.floor(ym)) * np.float32(32)) / 32.0).astype(np.float64)[..., None]
    p = lambda yy, xx: src[yy, xx].astype(np.float64)       # the window has a margin: no border access
    v = p(y0, x0) * (1 - ux) * (1 - uy) + p(y0, x0 + 1) * ux * (1 - uy) + p(y0 + 1, x0) * (1 - ux) * uy + p(y0 + 1, x0 + 1) * ux * uy
    return v


def test_oracle_reproduces_the_references_warped_artifact(oracle, art):
    xm, ym = _maps(oracle, art)
    src = art["src_window"]
    ref = art["artifact_crop"]
    g = float(art["gain"])
    assert xm.min() >= 1 and ym.min() >= 1 and xm.max() < src.shape[1] - 2 and ym.max() < src.shape[0] - 2
    exact = _tie_even_bilinear(src, xm, ym)
    half_even = np.clip(np.rint(exact), 0, 255).astype(np.uint8)                  # the OpenCL variant the artefact came from
    cpu = oracle.remap(src, xm, ym, oracle.LINEAR, oracle.BORDER_REFLECT)           # OpenCV's CPU fixed-point variant = this build
    # 1. the two variants are the same weighted sum; they differ exactly where it is a tie, and then by the rounding rule
    assert np.array_equal(cpu, np.clip(np.floor(exact + 0.5), 0, 255).astype(np.uint8))
    ties = (exact - np.floor(exact)) == 0.5
    assert np.array_equal(cpu != half_even, ties & (np.floor(exact) % 2 == 0))
    assert 0.005 < ties.mean() < 0.2                                                 # the crop holds the u = 0 band where fx = fy = 1/2
    # 2. with the artefact's rounding, warp + gain reproduce the committed bitmap
    out_even = oracle.gain_apply(half_even, g)
    d = out_even.astype(int) - ref
    assert (d != 0).mean() < 1e-3 and np.abs(d).max() <= 6        # 0.05 % here: the crop holds the sensitive column 799
    # 3. this build's (CPU-path) rounding differs from it only through those ties
    out_cpu = oracle.gain_apply(cpu, g)
    dc = out_cpu.astype(int) - ref
    assert (dc != 0).mean() < 0.06                                                   # 4 % in this crop (the tie band), 1.1 % over the whole tile
    explained = (cpu != half_even) | (d != 0)
    assert not ((dc != 0) & ~explained).any()


def test_gain_is_a_single_scalar(oracle, art):
    """GainCompensator (S:1165-1171): one gain per image — every pixel of the crop is consistent with the same value."""
    xm, ym = _maps(oracle, art)
    w = np.clip(np.rint(_tie_even_bilinear(art["src_window"], xm, ym)), 0, 255)
    ref = art["artifact_crop"].astype(np.float64)
    m = (w > 40) & (ref < 255)
    lo, hi = ((ref[m] - 0.5) / w[m]), ((ref[m] + 0.5) / w[m])
    g = float(art["gain"])
    assert (lo <= g).mean() > 0.999 and (hi >= g).mean() > 0.999


# ---- the reference's own DP seam -------------------------------------------------------------------------------------
def _seam_case(art):
    """Inputs of estimateSeam (S:806-957) for the window of the reference's committed seam: CV_32F images as the demo
    builds them (convertTo(CV_32F), S:1188-1190 / W:261), the crops standing in for the warped tiles (tile corner =
    crop corner), a rectangular component around the seam, tips = the seam's own end points."""
    img0 = art["img0"].astype(np.float32)
    img1 = art["img1"].astype(np.float32)
    tl = tuple(int(v) for v in art["crop_tl"])
    utl = tuple(int(v) for v in art["union_tl"])
    uw, uh = [int(v) for v in art["union_size"]]
    roi = tuple(int(v) for v in art["roi"])
    labels = np.zeros((uh, uw), np.int32)
    labels[roi[1]:roi[1] + roi[3], roi[0]:roi[0] + roi[2]] = 1
    seam = art["seam"]
    return dict(img1=img0, img2=img1, tl1=tl, tl2=tl, union_tl=utl, labels=labels, label=1, roi=roi,
                p1=tuple(int(v) for v in seam[0]), p2=tuple(int(v) for v in seam[-1])), seam


@pytest.fixture(scope="module")
def seam_art():
    return np.load(os.path.join(HERE, "golden", "ref_seam_artifact.npz"))


def test_oracle_reproduces_the_references_dp_seam(oracle, seam_art):
    """mask_seam[0].bmp / mask_seam[1].bmp (S:1197-1198) are the output of the in-tree DP seam finder on
    images_warped_f[0,1].bmp.  Their common boundary over 801 rows is the optimal path of estimateSeam's dynamic
    programme between its own end points: computeCosts + the programme + backtracking of the oracle return exactly it."""
    c, seam = _seam_case(seam_art)
    got, horiz = oracle.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p1"], c["p2"])
    assert not horiz and len(seam) == 801
    assert np.array_equal(got, seam)
    # the same seam from the other end (S:829-842 swaps the tips, S:947-948 restores the order)
    back, _ = oracle.seam_estimate(c["img1"], c["img2"], c["tl1"], c["tl2"], c["union_tl"], c["labels"], c["label"], c["roi"], c["p2"], c["p1"])
    assert np.array_equal(back, seam[::-1])


def test_oracle_reproduces_the_references_warped_mask(oracle, art):
    """mask_seam[0].bmp left of the second tile is the untouched warp of tile 0's all-255 mask (INTER_NEAREST,
    BORDER_CONSTANT, W:232 / S:1159): 799 x 1100 pixels, barrel-shaped boundary included, equal to the oracle's."""
    f, c = float(art["focal"]), float(art["centre"])
    K = np.array([[f, 0, c], [0, f, c], [0, 0, 1]], np.float32)
    w, h = [int(v) for v in art["src_size"]]
    corner, mk, _ = oracle.warp_u8(oracle.CYL, f, K, np.eye(3, dtype=np.float32), np.full((h, w), 255, np.uint8), oracle.NEAREST, oracle.BORDER_CONSTANT)
    assert corner == (int(art["roi"][0]), int(art["roi"][1]))
    n = int(art["mask_cols"])
    zr = art["mask_zero_rows"]
    assert np.array_equal(mk[zr, :n], art["mask_rows"])
    assert (np.delete(mk[:, :n], zr, axis=0) == 255).all()
    assert (art["mask_rows"] == 0).sum() > 3000          # the boundary is not trivial


# ---- the reference's whole DP seam finder run ------------------------------------------------------------------------
def dpseam_case(path=None):
    """Inputs / outputs of the reference's `find(images_warped_f, corners, masks_seam)` (S:1192) from
    tests/golden/ref_dpseam_artifact.npz: full-size CV_32FC3 images (zero outside the overlap strips — nothing else is
    read), corners, the masks that went in and the committed mask_seam[0,1].bmp that came out."""
    a = np.load(path or os.path.join(HERE, "golden", "ref_dpseam_artifact.npz"))
    s0, s1 = tuple(int(v) for v in a["shape0"]), tuple(int(v) for v in a["shape1"])
    i0, i1 = np.zeros(s0, np.float32), np.zeros(s1, np.float32)
    ow = a["strip0"].shape[1]
    i0[:, s0[1] - ow:] = a["strip0"]
    i1[:, :ow] = a["strip1"]
    unpack = lambda k, s: (np.unpackbits(a[k], axis=1)[:, :s[1]] * 255).astype(np.uint8)
    return dict(images=[i0, i1], corners=[tuple(int(v) for v in a["tl0"]), tuple(int(v) for v in a["tl1"])],
                masks_in=[unpack("mask_in0", s0), unpack("mask_in1", s1)], masks_out=[unpack("mask_out0", s0), unpack("mask_out1", s1)])


def test_oracle_reproduces_the_references_seam_masks():
    """The whole in-tree DP seam finder (S:87-1093) restated in oracle/dpseam_np.py turns the masks that went in into
    exactly the reference's committed mask_seam[0].bmp and mask_seam[1].bmp — every one of the 1086 x 1100 and
    1096 x 1102 pixels."""
    from oracle.dpseam_np import DpSeamFinder
    c = dpseam_case()
    masks = [m.copy() for m in c["masks_in"]]
    DpSeamFinder().find(c["images"], c["corners"], masks)
    assert np.array_equal(masks[0], c["masks_out"][0]) and np.array_equal(masks[1], c["masks_out"][1])
    assert (masks[0] != c["masks_in"][0]).sum() > 100000 and (masks[1] != c["masks_in"][1]).sum() > 50000      # it did cut both


def _demo_blend(feather, dilate_rect, gain_unused, c, sharpness, dilate):
    """S:1236-1283: FeatherBlender(sharpness), dilate(masks_seam, 20x20) & masks_warped, convertTo(CV_16S), feed, blend."""
    sizes = [(im.shape[1], im.shape[0]) for im in c["images"]]
    fb = feather(sharpness)
    fb.prepare(c["corners"], sizes)
    for img, ms, mw, tl in zip(c["images"], c["masks_out"], c["masks_in"], c["corners"]):
        mk = (dilate_rect(ms, 20, 20) & mw) if dilate else ms
        fb.feed(img.astype(np.int16), mk, tl)
    res, rm = fb.blend()
    return np.clip(np.asarray(res), 0, 255).astype(np.int32), np.asarray(rm)


def _psnr_in_zone(res, rm, a):
    x0, x1 = [int(v) for v in a["pano_zone_x"]]
    err = (res[:, x0:x1] - a["pano_zone"].astype(np.int32)).astype(np.float64)
    valid = rm[:, x0:x1] > 0
    return 10 * np.log10(255.0 ** 2 / (err[valid] ** 2).mean())


def test_feather_stage_agrees_with_the_references_pano(oracle):
    """pano.jpg (S:1283) is the FeatherBlender(0.1) result of exactly the artefacts above.  It is JPEG-compressed, so the
    comparison is a PSNR — but a discriminating one: in the 130-column band around the seam, where the output is all blend,
    the oracle's stage (dilate 20x20 & mask, distance-transform weights with sharpness 0.1, normalise) is as close to the
    JPEG as JPEG noise allows (> 42 dB, like the untouched areas), while leaving out the dilation or using OpenCV's
    default sharpness 0.02 is off by 6-13 dB."""
    a = np.load(os.path.join(HERE, "golden", "ref_dpseam_artifact.npz"))
    c = dpseam_case()
    feather = lambda s: oracle.Feather(s)
    ref = _psnr_in_zone(*_demo_blend(feather, oracle.dilate_rect, None, c, 0.1, True), a)
    no_dilate = _psnr_in_zone(*_demo_blend(feather, oracle.dilate_rect, None, c, 0.1, False), a)
    soft = _psnr_in_zone(*_demo_blend(feather, oracle.dilate_rect, None, c, 0.02, True), a)
    assert ref > 42.0 and no_dilate < ref - 8.0 and soft < ref - 4.0, (ref, no_dilate, soft)


def _maps_artifact():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_maps_artifact.npz"))


def check_maps_against_the_reference_bitmaps(xm, ym):
    """saturate_cast<uchar>(cvRound(map)) - what imwrite("xmap.bmp", xmap) stored (W:155-156) - against the committed bitmaps: every one of
    the 2 x 1102 x 1096 values, except a handful that sit within 1e-4 of a rounding tie (K and R of the author's run are recovered by a
    fit, tests/golden/make_golden_maps.py, so the last bits of K * R^T are not the author's)."""
    D = _maps_artifact()
    total = 0
    for m, ref in ((xm, D["xmap_u8"]), (ym, D["ymap_u8"])):
        assert m.shape == ref.shape == (1102, 1096)
        got = np.clip(np.rint(m), 0, 255).astype(np.int16)
        bad = got != ref.astype(np.int16)
        total += int(bad.sum())
        assert np.abs(got - ref.astype(np.int16)).max() <= 1
        tie = np.abs((m - np.floor(m)) - 0.5)[bad]
        assert tie.size == 0 or tie.max() < 1e-4, tie.max()
        # the saturated parts too: negative coordinates -> 0, coordinates beyond 255 -> 255
        assert (ref[(m < -0.5)] == 0).all() and (ref[m > 255.5] == 255).all()
    assert total <= 20, total
    return total


def test_buildmaps_reproduces_the_references_xmap_ymap_bitmaps(oracle):
    """xmap.bmp / ymap.bmp of the reference (W:155-156: the maps of its last warp() call): setCameraParams (W:90-120: K * R^T), the map
    fill of buildMaps with mapBackward (W:46-63, 133-141) and the 8-bit conversion reproduce both bitmaps with the camera recovered from
    them - f = 2707.68 for a scale of 2707.47 (W:30), principal point 550.5 as for the 1101 x 1101 sources."""
    D = _maps_artifact()
    K, R, tl = D["K"], D["R"], D["tl"]
    assert abs(float(K[0, 0]) - 2707.68) < 0.05 and K[0, 2] == 550.5 and K[1, 2] == 550.5
    _, _, _, kr = oracle.camera(K, R)
    roi = [int(tl[0]), int(tl[1]), int(tl[0]) + 1096 - 1, int(tl[1]) + 1102 - 1]
    xm, ym = oracle.build_maps(oracle.CYL, float(D["scale"]), kr, roi)
    n = check_maps_against_the_reference_bitmaps(xm, ym)
    assert n == 14   # 6 + 8 values at rounding ties


def test_costv_support_matches_the_references_costv_bitmap(oracle):
    """costV.bmp (B:265) is the SSD cost map of the in-tree pair blend (B:207-261) for a pair whose inputs are not in the reference tree;
    what it does record is the map's shape, interSectHe_ x (interSectBr_ + 2) = 1105 x 284, and its support: columns 1 .. interSectBr_ - 2
    (the loop B:223), rows dy .. interSectHe_ - dy.  A pair of that geometry through the restatement has exactly that shape and column
    support (A13's cost term; the greedy seam B:268-307 reads this array)."""
    D = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_costv_artifact.npz"))
    He, cw = [int(v) for v in D["shape"]]
    assert (He, cw) == (1105, 284)
    ibr = cw - 2
    col_any = D["col_any"]
    assert not col_any[0] and not col_any[ibr - 1:].any() and col_any[1:ibr - 1].all()        # x in [1, interSectBr_ - 1)
    rng = np.random.default_rng(5)
    dy = 5
    h1, w1, h2, w2 = He - dy, 1086, He - dy, 1096                                          # sizes of the kind the demos produce
    img1 = rng.integers(1, 255, (h1, w1, 3)).astype(np.float32)
    img2 = rng.integers(1, 255, (h2, w2, 3)).astype(np.float32)
    tl1, tl2 = (0, 0), (w1 - ibr, dy)
    rc, cost = oracle.pair_linear_costv(img1, img2, tl1, tl2)
    assert rc == 0 and cost.shape == (He, cw)
    assert np.array_equal((cost != 0).any(0), col_any)
    rows = (cost != 0).any(1)
    assert not rows[:dy].any() and not rows[He - dy:].any() and rows[dy:He - dy].all()
    # the artefact's zero rows at the top and bottom are at least the dy band (it also holds rows where both tiles are black)
    assert not D["row_any"][:dy].any() and not D["row_any"][He - dy:].any()
