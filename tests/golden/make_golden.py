#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the BUILD container only, where
/root/reference exists; the GPU box and the tests only read the .npz files).

  cyl_maps.npz       inputs + outputs of the reference's OWN mapForward / mapBackward (W:30-63 compiled
                     verbatim into oracle/_ref/libref_warp.so by oracle/build.sh).  This is the one
                     part of the hot path the reference holds in compilable form; it pins
                     oracle.c's projector bit-exactly.
  ref_inputs.npz     data crops of the reference's committed artefacts (mask_seam[0,1].bmp,
                     images_warped_f[0,1].bmp written by S:1195-1198): realistic warped tiles and real
                     DP-seam masks used as blend INPUTS.  Data files, not source.
  ref_warp_artifact.npz  the one artefact of the reference whose inputs can be reconstructed: images_warped_f[0].bmp
                     (S:1195) turns out to be warp(src2.bmp) with f = 2707.47 (W:30), c = 550.5, R = I, times the
                     GainCompensator's gain — a crop of it with the source window it is sampled from.
  ref_seam_artifact.npz  the reference's own DP seam (the boundary between its committed mask_seam[0].bmp and
                     mask_seam[1].bmp, S:1197-1198) over 801 rows, with the crops of images_warped_f[0,1].bmp the
                     cost maps are computed from.  The tile offset (799, -5) is the one for which the two seam masks
                     partition the overlap exactly (and gives pano.jpg's 1895 x 1105 union).
  ref_dpseam_artifact.npz  inputs and outputs of the reference's whole DP seam finder run (S:1192 `find`): the overlap strips of
                     images_warped_f[0,1].bmp (the only pixels its cost maps read), the warped masks that went in (tile 0:
                     the warp of an all-255 mask; tile 1: reconstructed — its part outside the overlap is untouched in
                     mask_seam[1].bmp, inside the overlap every pixel ended up in one of the two seam masks) and the
                     committed mask_seam[0,1].bmp that came out.
  oracle_regress.npz seeded inputs -> outputs of oracle/liboracle.so for remap / pyramids /
                     MultiBandBlender / linear blend.  NOT reference-derived (OpenCV 3.4.2 is absent:
                     "parity unpinned"); it freezes the restatement so that drift is caught.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import capi as O  # noqa: E402
from imagestitch_amd import synth  # noqa: E402

REF = "/root/reference"


def cyl_maps():
    assert O.ref() is not None, "oracle/_ref/libref_warp.so missing: run oracle/build.sh in the build container"
    rng = np.random.default_rng(20260928)
    out = {}
    cams = []
    # the reference's own geometry: scale = 2707.47f (W:30), 1101 x 1101 source (src1.bmp), c = 550.5
    f = np.float32(2707.47)
    cams.append((f, np.array([[f, 0, 550.5], [0, f, 550.5], [0, 0, 1]], np.float32), np.eye(3, dtype=np.float32), 1101, 1101))
    for (w, h, fo, yaw) in ((3840, 2160, 3000.0, 0.36), (1920, 1080, 1500.0, -0.3), (640, 480, 300.0, 0.9)):
        K, Rs = synth.camera_pair(w, h, fo, yaw=abs(yaw))
        cams.append((np.float32(fo), K, Rs[0] if yaw > 0 else Rs[1], w, h))
    for i, (scale, K, R, w, h) in enumerate(cams):
        k, rinv, r_kinv, k_rinv = O.camera(K, R)
        O.ref_set(float(scale), r_kinv, k_rinv)
        # forward: image corners, borders and random interior points (integer pixel coordinates, as W:76 casts)
        xs = np.concatenate([[0, w - 1, 0, w - 1], rng.integers(0, w, 400), np.arange(0, w, max(w // 64, 1))]).astype(np.float32)
        ys = np.concatenate([[0, 0, h - 1, h - 1], rng.integers(0, h, 400), np.zeros(len(np.arange(0, w, max(w // 64, 1))))]).astype(np.float32)
        u, v = O.ref_map_forward_n(xs, ys)
        # backward: integer (u, v) across and beyond the ROI (z <= 0 cases included for wide yaw)
        uu = rng.integers(int(u.min()) - 50, int(u.max()) + 50, 600).astype(np.float32)
        vv = rng.integers(int(v.min()) - 50, int(v.max()) + 50, 600).astype(np.float32)
        uu[:4] = [scale * 1.7, -scale * 1.7, scale * 3.2, 0]     # beyond +-pi/2: z <= 0 -> (-1,-1)
        x, y = O.ref_map_backward_n(uu, vv)
        out.update({"scale%d" % i: np.float32(scale), "r_kinv%d" % i: r_kinv, "k_rinv%d" % i: k_rinv, "size%d" % i: np.array([w, h]),
                    "fx%d" % i: xs, "fy%d" % i: ys, "fu%d" % i: u, "fv%d" % i: v, "bu%d" % i: uu, "bv%d" % i: vv, "bx%d" % i: x, "by%d" % i: y})
    out["n"] = np.array(len(cams))
    np.savez_compressed(os.path.join(HERE, "cyl_maps.npz"), **out)
    print("cyl_maps.npz:", len(cams), "cameras")


def ref_inputs():
    from PIL import Image
    d = os.path.join(REF, "动态规划法寻找最佳缝合线", "动态规划法寻找最佳缝合线")
    m0 = np.array(Image.open(os.path.join(d, "mask_seam[0].bmp")).convert("L"))
    m1 = np.array(Image.open(os.path.join(d, "mask_seam[1].bmp")).convert("L"))
    i0 = np.array(Image.open(os.path.join(d, "images_warped_f[0].bmp")).convert("RGB"))[:, :, ::-1]  # BGR as imread gives
    i1 = np.array(Image.open(os.path.join(d, "images_warped_f[1].bmp")).convert("RGB"))[:, :, ::-1]
    print("reference artefact shapes:", m0.shape, m1.shape, i0.shape, i1.shape)
    # blend INPUTS only: any offset gives a valid pair of overlapping crops.  (The offset of the reference's own run is
    # (799, -5): see ref_seam_artifact / ref_dpseam_artifact below.)
    dx, dy = 799, 3
    # crop a 448 x 320 window of each tile around the seam (panorama x in [760, 1208), y in [400, 720))
    px0, py0, cw, ch = 760, 400, 448, 320
    c0 = (slice(py0, py0 + ch), slice(px0, px0 + cw))
    c1 = (slice(py0 - dy, py0 - dy + ch), slice(max(px0 - dx, 0), px0 - dx + cw))
    np.savez_compressed(os.path.join(HERE, "ref_inputs.npz"),
                        img0=np.ascontiguousarray(i0[c0]), img1=np.ascontiguousarray(i1[c1]),
                        mask0=np.ascontiguousarray(m0[c0]), mask1=np.ascontiguousarray(m1[c1]),
                        corner0=np.array([px0, py0]), corner1=np.array([max(px0, dx), py0]),
                        full_shapes=np.array([m0.shape, m1.shape]))
    print("ref_inputs.npz: crops", i0[c0].shape, i1[c1].shape)


def ref_warp_artifact():
    from PIL import Image
    src = np.array(Image.open(os.path.join(REF, "特征点检测", "特征点检测", "src2.bmp")).convert("RGB"))[:, :, ::-1]
    art = np.array(Image.open(os.path.join(REF, "动态规划法寻找最佳缝合线", "动态规划法寻找最佳缝合线", "images_warped_f[0].bmp")).convert("RGB"))[:, :, ::-1]
    f = np.float32(2707.47)
    K = np.array([[f, 0, 550.5], [0, f, 550.5], [0, 0, 1]], np.float32)
    R = np.eye(3, dtype=np.float32)
    roi, _ = O.detect_roi(O.CYL, float(f), K, R, src.shape[1], src.shape[0])
    assert (roi[2] - roi[0] + 1, roi[3] - roi[1] + 1) == (art.shape[1], art.shape[0]), (roi, art.shape)
    oy, ox, oh, ow = 430, 470, 300, 370                     # output crop: the centre band (u = 0 at column 543) and column 799
    k, rinv, r_kinv, k_rinv = O.camera(K, R)
    sub = np.array([roi[0] + ox, roi[1] + oy, roi[0] + ox + ow - 1, roi[1] + oy + oh - 1], np.int32)
    xm, ym = O.build_maps(O.CYL, float(f), k_rinv, sub)
    sx0, sy0 = int(np.floor(xm.min())) - 2, int(np.floor(ym.min())) - 2
    sx1, sy1 = int(np.ceil(xm.max())) + 3, int(np.ceil(ym.max())) + 3
    assert sx0 >= 0 and sy0 >= 0 and sx1 <= src.shape[1] and sy1 <= src.shape[0]
    # mask_seam[0].bmp left of the second tile (x < 799) is the untouched warped mask of tile 0 (W:232 / S:1159: all-255 mask,
    # INTER_NEAREST, BORDER_CONSTANT): its barrel-shaped boundary pins the nearest-neighbour path.  Rows that hold a zero:
    ms = np.array(Image.open(os.path.join(REF, "动态规划法寻找最佳缝合线", "动态规划法寻找最佳缝合线", "mask_seam[0].bmp")).convert("L"))[:, :799]
    zrows = np.nonzero((ms == 0).any(1))[0]
    assert (np.delete(ms, zrows, axis=0) == 255).all()
    np.savez_compressed(os.path.join(HERE, "ref_warp_artifact.npz"), mask_cols=np.array(799), mask_zero_rows=zrows, mask_rows=np.ascontiguousarray(ms[zrows]),
                        src_window=np.ascontiguousarray(src[sy0:sy1, sx0:sx1]), src_origin=np.array([sx0, sy0]),
                        src_size=np.array([src.shape[1], src.shape[0]]), artifact_crop=np.ascontiguousarray(art[oy:oy + oh, ox:ox + ow]),
                        crop_origin=np.array([ox, oy]), artifact_size=np.array([art.shape[1], art.shape[0]]), focal=f, centre=np.float32(550.5),
                        roi=roi, gain=np.float64(0.988722))
    print("ref_warp_artifact.npz: source window", (sy1 - sy0, sx1 - sx0), "artefact crop", (oh, ow))


def ref_seam_artifact():
    from PIL import Image
    d = os.path.join(REF, "动态规划法寻找最佳缝合线", "动态规划法寻找最佳缝合线")
    m0 = np.array(Image.open(os.path.join(d, "mask_seam[0].bmp")).convert("L"))
    m1 = np.array(Image.open(os.path.join(d, "mask_seam[1].bmp")).convert("L"))
    i0 = np.array(Image.open(os.path.join(d, "images_warped_f[0].bmp")).convert("RGB"))[:, :, ::-1]
    i1 = np.array(Image.open(os.path.join(d, "images_warped_f[1].bmp")).convert("RGB"))[:, :, ::-1]
    tl0, tl1 = (-543, -550), (-543 + 799, -550 - 5)
    utl = (min(tl0[0], tl1[0]), min(tl0[1], tl1[1]))
    o0, o1 = (tl0[0] - utl[0], tl0[1] - utl[1]), (tl1[0] - utl[0], tl1[1] - utl[1])
    uw = max(o0[0] + m0.shape[1], o1[0] + m1.shape[1])
    uh = max(o0[1] + m0.shape[0], o1[1] + m1.shape[0])
    U0 = np.zeros((uh, uw), np.uint8); U1 = np.zeros((uh, uw), np.uint8)
    U0[o0[1]:o0[1] + m0.shape[0], o0[0]:o0[0] + m0.shape[1]] = m0
    U1[o1[1]:o1[1] + m1.shape[0], o1[0]:o1[0] + m1.shape[1]] = m1
    assert not ((U0 > 0) & (U1 > 0)).any()                      # the two seam masks never overlap with this offset
    ya, yb = 150, 950
    seam = []
    for y in range(ya, yb + 1):   # vertical seam: the seam pixel is the first pixel of the right-hand tile (S:1045-1052)
        a = np.nonzero(U0[y, o1[0]:o0[0] + m0.shape[1]])[0]
        b = np.nonzero(U1[y, o1[0]:o0[0] + m0.shape[1]])[0]
        assert b.min() == a.max() + 1
        seam.append((o1[0] + int(b.min()), y))
    seam = np.array(seam, np.int32)
    xa, xb = int(seam[:, 0].min()) - 36, int(seam[:, 0].max()) + 36
    # image crops (union rows ya - 2 .. yb + 2, columns xa - 2 .. xb + 2) and their corners in panorama coordinates
    cy0, cy1, cx0, cx1 = ya - 2, yb + 3, xa - 2, xb + 3
    c0 = i0[cy0 - o0[1]:cy1 - o0[1], cx0 - o0[0]:cx1 - o0[0]]
    c1 = i1[cy0 - o1[1]:cy1 - o1[1], cx0 - o1[0]:cx1 - o1[0]]
    assert c0.shape == c1.shape == (cy1 - cy0, cx1 - cx0, 3)
    np.savez_compressed(os.path.join(HERE, "ref_seam_artifact.npz"), img0=np.ascontiguousarray(c0), img1=np.ascontiguousarray(c1),
                        crop_tl=np.array([utl[0] + cx0, utl[1] + cy0]), union_tl=np.array(utl), roi=np.array([xa, ya, xb - xa, yb - ya + 1]),
                        seam=seam, tile_offset=np.array([799, -5]), union_size=np.array([uw, uh]))
    print("ref_seam_artifact.npz: seam rows", ya, yb, "roi x", xa, xb, "crops", c0.shape)


def ref_dpseam_artifact():
    from PIL import Image
    d = os.path.join(REF, "动态规划法寻找最佳缝合线", "动态规划法寻找最佳缝合线")
    m0 = np.array(Image.open(os.path.join(d, "mask_seam[0].bmp")).convert("L"))
    m1 = np.array(Image.open(os.path.join(d, "mask_seam[1].bmp")).convert("L"))
    i0 = np.array(Image.open(os.path.join(d, "images_warped_f[0].bmp")).convert("RGB"))[:, :, ::-1]
    i1 = np.array(Image.open(os.path.join(d, "images_warped_f[1].bmp")).convert("RGB"))[:, :, ::-1]
    dx, dy = 799, -5
    f = np.float32(2707.47)
    K = np.array([[f, 0, 550.5], [0, f, 550.5], [0, 0, 1]], np.float32)
    _, mk0, _ = O.warp_u8(O.CYL, float(f), K, np.eye(3, dtype=np.float32), np.full((1101, 1101), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
    ow = m0.shape[1] - dx                                       # overlap width: tile-0 columns [dx, W0) = tile-1 columns [0, ow)
    mk1 = m1.copy()
    for y1 in range(m1.shape[0]):
        y0 = y1 + dy
        if 0 <= y0 < m0.shape[0]:
            mk1[y1, :ow] = np.where((m0[y0, dx:] > 0) | (m1[y1, :ow] > 0), 255, 0)
    # the seam band of the demo's final result, pano.jpg (S:1283: FeatherBlender 0.1 on these very inputs, JPEG-compressed)
    pano = np.array(Image.open(os.path.join(d, "pano.jpg")).convert("RGB"))[:, :, ::-1]
    assert pano.shape[:2] == (1105, 1895)
    zx0, zx1 = 880, 1010
    np.savez_compressed(os.path.join(HERE, "ref_dpseam_artifact.npz"), pano_zone=np.ascontiguousarray(pano[:, zx0:zx1]), pano_zone_x=np.array([zx0, zx1]), strip0=np.ascontiguousarray(i0[:, dx:]), strip1=np.ascontiguousarray(i1[:, :ow]),
                        shape0=np.array(i0.shape), shape1=np.array(i1.shape), tl0=np.array([-543, -550]), tl1=np.array([-543 + dx, -550 + dy]),
                        mask_in0=np.packbits(mk0 > 0, axis=1), mask_in1=np.packbits(mk1 > 0, axis=1),
                        mask_out0=np.packbits(m0 > 0, axis=1), mask_out1=np.packbits(m1 > 0, axis=1))
    print("ref_dpseam_artifact.npz: overlap strips", i0[:, dx:].shape, i1[:, :ow].shape)


def oracle_regress():
    rng = np.random.default_rng(7)
    out = {}
    src = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    xm = (rng.random((33, 47)) * 70 - 7).astype(np.float32)
    ym = (rng.random((33, 47)) * 52 - 6).astype(np.float32)
    out.update(remap_src=src, remap_x=xm, remap_y=ym)
    for interp, border in ((1, 2), (0, 0), (1, 0), (1, 4)):
        out["remap_%d_%d" % (interp, border)] = O.remap(src, xm, ym, interp, border)
    a = rng.integers(-4000, 4000, (22, 30, 3)).astype(np.int16)
    f = (rng.standard_normal((22, 30, 3)) * 90).astype(np.float32)
    out.update(pyr_s16=a, pyr_f32=f, down_s16=O.pyr_down(a), up_s16=O.pyr_up(a), down_f32=O.pyr_down(f), up_f32=O.pyr_up(f))
    corners, sizes = [(-5, 3), (40, -2)], [(70, 50), (64, 57)]
    tiles = [(rng.integers(0, 256, (s[1], s[0], 3)).astype(np.int16), (rng.random((s[1], s[0])) > 0.3).astype(np.uint8) * 255) for s in sizes]
    out.update(mb_corners=np.array(corners), mb_sizes=np.array(sizes), mb_img0=tiles[0][0], mb_mask0=tiles[0][1], mb_img1=tiles[1][0], mb_mask1=tiles[1][1])
    for prec in (0, 1, 2):
        mb = O.MultiBand(4, prec)
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            mb.feed(img, mask, c)
        d, m = mb.blend(prec != 0)
        out["mb_dst%d" % prec] = d
        out["mb_omask%d" % prec] = m
    i1 = (rng.random((60, 90, 3)) * 255).astype(np.float32)
    i2 = (rng.random((63, 80, 3)) * 255).astype(np.float32)
    i1[:8, -12:] = 2.0
    i2[-9:, :10] = 1.0
    rc, pano, seam = O.blend_pair_linear(i1, i2, (5, 9), (5 + 50, 9 + 2))
    assert rc == 0
    out.update(lin_img1=i1, lin_img2=i2, lin_pano=pano, lin_seam=seam)
    np.savez_compressed(os.path.join(HERE, "oracle_regress.npz"), **out)
    print("oracle_regress.npz written")


if __name__ == "__main__":
    cyl_maps()
    if os.path.isdir(REF):
        ref_inputs()
        ref_warp_artifact()
        ref_seam_artifact()
        ref_dpseam_artifact()
    oracle_regress()
