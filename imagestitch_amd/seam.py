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
