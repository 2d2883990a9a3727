"""The data-parallel part of the reference's in-tree DP seam finder: estimateSeam S:806-957 with
computeCosts S:733-803 — cost maps and dynamic programme on the GPU.  The component analysis around it stays with the caller."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_mat, check


def seam_estimate(image1, image2, tl1, tl2, union_tl, labels, label, roi, p1, p2, device=0, stream=None):
    """estimateSeam(image1, image2, tl1, tl2, comp, p1, p2, seam, isHorizontal).  labels = labels_ (HxW int32, union-sized),
    label = comp + 1, roi = (x, y, width, height) of Rect(tls_[comp], brs_[comp]); points are (x, y) in union coordinates.
    Returns (seam as an (N, 2) int32 array, p1 first — empty when p2 is not reachable —, is_horizontal)."""
    m1, m2, ml = as_mat(image1), as_mat(image2), as_mat(labels)
    cap = int(roi[2]) + int(roi[3]) + 2
    out = np.zeros((cap, 2), np.int32)
    n, horiz = C.c_int(0), C.c_int(0)
    r = (C.c_int * 4)(*[int(v) for v in roi])
    ptr = getattr(stream, "cuda_stream", stream)
    check(_lib.load().isx_seam_estimate(C.byref(m1), C.byref(m2), int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]), int(union_tl[0]), int(union_tl[1]),
                                        C.byref(ml), int(label), r, int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1]),
                                        out.ctypes.data_as(C.POINTER(C.c_int)), cap, C.byref(n), C.byref(horiz), int(device), C.c_void_p(ptr or 0)))
    return out[: n.value].copy(), bool(horiz.value)


class DpSeamFinder:
    """The reference's in-tree DP seam finder (S:60-1093, `find` as called at S:1192; costFunc_ COLOR)."""

    def __init__(self, device=0, stream=None):
        self.device, self.stream = device, stream

    def find(self, src, corners, masks):
        """find(src, corners, masks): src = CV_32FC3 (or CV_8UC3) images, masks = CV_8U arrays / tensors edited in place."""
        n = len(src)
        mats_i = (_lib.IsxMat * n)(*[as_mat(a) for a in src])
        mats_m = (_lib.IsxMat * n)(*[as_mat(m) for m in masks])
        c = (C.c_int * (2 * n))(*[int(v) for p in corners for v in p])
        ptr = getattr(self.stream, "cuda_stream", self.stream)
        check(_lib.load().isx_dp_seam_find(n, mats_i, c, mats_m, int(self.device), C.c_void_p(ptr or 0)))
        return masks

    @staticmethod
    def release():
        """Return the work images the finder keeps per calling thread between find() calls (isx_dp_seam_release)."""
        check(_lib.load().isx_dp_seam_release())
