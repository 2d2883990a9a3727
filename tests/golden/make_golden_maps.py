#!/usr/bin/env python3
"""Generates tests/golden/ref_maps_artifact.npz from the reference's committed xmap.bmp / ymap.bmp (run in the build container, where
/root/reference exists; the GPU box only sees the .npz).

The two bitmaps are what W:155-156 wrote: imwrite("xmap.bmp", xmap) / imwrite("ymap.bmp", ymap) of the CV_32F maps of the LAST warp()
call of the author's run (the mask of image 1, W:232), i.e. saturate_cast<uchar>(cvRound(map)) - 1102 x 1096 values each, of which the
first ~260 columns (xmap) / rows (ymap) are not saturated.  K and R of that call were never recorded.  This script recovers them:
  K = [f 0 550.5; 0 f 550.5; 0 0 1]   (HomographyBasedEstimator / BundleAdjusterRay: focal and rotation per camera, principal point =
                                       image centre of the 1101 x 1101 sources), R = rotation vector r, scale = 2707.47f (W:30),
  map corner tl = (-917, -555)         (the corner's x is degenerate with the rotation about the vertical axis: any integer works)
by least squares on the unsaturated values followed by a hinge refinement (every value must lie within 1/2 of its rounded value): the
refined model violates the 1/2 bound in 82 of 549 759 unsaturated values, by at most 1e-4 - the evaluation noise of fp32.
The .npz holds the two bitmaps as they are (data, zlib-compressed) and the fitted K, R (float32).  tests/test_ref_artifact.py and
tests/test_gpu_warp.py then demand that setCameraParams + buildMaps + the 8-bit conversion reproduce the bitmaps, all 2.4 M values, up to
the handful (14) that sit within 1e-4 of a rounding tie."""
import os
import sys

import numpy as np
from PIL import Image
from scipy.optimize import least_squares

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = "/root/reference/圆柱面投影变换/圆柱面投影变换/"
X = np.array(Image.open(D + "xmap.bmp")).astype(np.float64)
Y = np.array(Image.open(D + "ymap.bmp")).astype(np.float64)
H, W = X.shape
S = np.float64(np.float32(2707.47))
TL = np.array([-917.0, -555.0])
ux, uy = (X > 0) & (X < 255), (Y > 0) & (Y < 255)
ix, iy = np.argwhere(ux), np.argwhere(uy)
rng = np.random.default_rng(0)
sx, sy = ix[rng.choice(len(ix), 60000, replace=False)], iy[rng.choice(len(iy), 60000, replace=False)]


def rot(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * kx + (1 - np.cos(th)) * kx @ kx


def maps(q, pts):
    K = np.array([[q[0], 0, 550.5], [0, q[0], 550.5], [0, 0, 1.0]])
    kr = K @ rot(q[1:4]).T
    u, v = (pts[:, 1] + TL[0]) / S, (pts[:, 0] + TL[1]) / S
    x_, y_, z_ = np.sin(u), v, np.cos(u)
    z = kr[2, 0] * x_ + kr[2, 1] * y_ + kr[2, 2] * z_
    return (kr[0, 0] * x_ + kr[0, 1] * y_ + kr[0, 2] * z_) / z, (kr[1, 0] * x_ + kr[1, 1] * y_ + kr[1, 2] * z_) / z


def res(q):
    return np.concatenate([maps(q, sx)[0] - X[sx[:, 0], sx[:, 1]], maps(q, sy)[1] - Y[sy[:, 0], sy[:, 1]]])


def hinge(q):
    e = np.concatenate([maps(q, ix)[0] - X[ix[:, 0], ix[:, 1]], maps(q, iy)[1] - Y[iy[:, 0], iy[:, 1]]])
    return np.maximum(np.abs(e) - 0.4999, 0.0)


r1 = least_squares(res, np.array([2707.3, 0.0, -0.136, 0.0]), x_scale=[10, 1e-3, 1e-3, 1e-3])
r2 = least_squares(hinge, r1.x, x_scale=[10, 1e-3, 1e-3, 1e-3], ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200)
q = r2.x
h = hinge(q)
print("least squares rms %.4f (rounding noise: %.4f); after the hinge refinement %d of %d values violate the 1/2 bound, by at most %.1e" %
      (np.sqrt(np.mean(r1.fun ** 2)), 1 / np.sqrt(12), int((h > 0).sum()), h.size, h.max()))
K = np.array([[q[0], 0, 550.5], [0, q[0], 550.5], [0, 0, 1.0]], np.float32)
R = rot(q[1:4]).astype(np.float32)
out = os.path.join(ROOT, "tests", "golden", "ref_maps_artifact.npz")
np.savez_compressed(out, xmap_u8=X.astype(np.uint8), ymap_u8=Y.astype(np.uint8), K=K, R=R, tl=TL.astype(np.int32), scale=np.float32(2707.47))
print("wrote", out, os.path.getsize(out), "bytes; f = %.4f" % q[0])
sys.path.insert(0, ROOT)
from oracle import capi as O  # noqa: E402
_, _, _, kr = O.camera(K, R)
roi = [int(TL[0]), int(TL[1]), int(TL[0]) + W - 1, int(TL[1]) + H - 1]
xm, ym = O.build_maps(O.CYL, float(np.float32(2707.47)), kr, roi)
dx, dy = np.clip(np.rint(xm), 0, 255) - X, np.clip(np.rint(ym), 0, 255) - Y
print("oracle maps vs the bitmaps: %d + %d of %d values differ" % ((dx != 0).sum(), (dy != 0).sum(), 2 * dx.size))

# ---- costV.bmp (B:265): its inputs (-22.bmp / -11.bmp) are not in the reference tree, so only its shape and support can be kept
CV = np.array(Image.open("/root/reference/图像融合/图像融合/costV.bmp"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_costv_artifact.npz"), shape=np.array(CV.shape, np.int32), col_any=(CV != 0).any(0),
                    row_any=(CV != 0).any(1), frac_saturated=np.float32((CV == 255).mean()))
