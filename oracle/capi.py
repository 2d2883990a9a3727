"""ctypes binding of oracle/liboracle.so (and oracle/_ref/libref_warp.so when present).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under imagestitch_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

CYL, SPH = 0, 1
NEAREST, LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
I16, F32, F16ACC32 = 0, 1, 2

_f9 = C.c_float * 9
_i4 = C.c_int * 4
_f4 = C.c_float * 4


def build():
    subprocess.check_call(["bash", os.path.join(_HERE, "build.sh")])


def lib():
    global _LIB
    if _LIB is None:
        # ISX_ORACLE_LIB: bench.py's cpu_baseline leg times a build of the same oracle.c made on the box it runs on
        # (-O3 -march=native); the parity tests always use the portable -O2 build
        path = os.environ.get("ISX_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_cvround.restype = C.c_int
        L.orc_cvround.argtypes = [C.c_float]
        L.orc_f2i_trunc.restype = C.c_int
        L.orc_f2i_trunc.argtypes = [C.c_float]
        L.orc_border_interpolate.restype = C.c_int
        L.orc_border_interpolate.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_f16_round.restype = C.c_float
        L.orc_f16_round.argtypes = [C.c_float]
        L.orc_mb_create.restype = C.c_void_p
        L.orc_mb_create.argtypes = [C.c_int, C.c_int]
        L.orc_mb_destroy.argtypes = [C.c_void_p]
        L.orc_mb_prepare.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_mb_num_bands.restype = C.c_int
        L.orc_mb_num_bands.argtypes = [C.c_void_p]
        L.orc_mb_result_size.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mb_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_mb_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mb_blend.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_blend_pair_linear.restype = C.c_int
        _LIB = L
    return _LIB


def ref():
    """The verbatim reference projector (W:30-63) or None when it was not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_warp.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# ---------------------------------------------------------------- geometry
def camera(K, R):
    K = _c(K, np.float32).reshape(9)
    R = _c(R, np.float32).reshape(9)
    k, rinv, r_kinv, k_rinv = (np.zeros(9, np.float32) for _ in range(4))
    lib().orc_camera(_p(K), _p(R), _p(k), _p(rinv), _p(r_kinv), _p(k_rinv))
    return k, rinv, r_kinv, k_rinv


def map_forward(kind, scale, r_kinv, x, y):
    u, v = C.c_float(), C.c_float()
    r = _c(r_kinv, np.float32)
    lib().orc_map_forward(C.c_int(kind), C.c_float(scale), _p(r), C.c_float(x), C.c_float(y), C.byref(u), C.byref(v))
    return np.float32(u.value), np.float32(v.value)


def map_backward(kind, scale, k_rinv, u, v):
    x, y = C.c_float(), C.c_float()
    r = _c(k_rinv, np.float32)
    lib().orc_map_backward(C.c_int(kind), C.c_float(scale), _p(r), C.c_float(u), C.c_float(v), C.byref(x), C.byref(y))
    return np.float32(x.value), np.float32(y.value)


def detect_roi(kind, scale, K, R, src_w, src_h):
    k, rinv, r_kinv, _ = camera(K, R)
    roi = np.zeros(4, np.int32)
    mm = np.zeros(4, np.float32)
    lib().orc_detect_roi(C.c_int(kind), C.c_float(scale), _p(k), _p(rinv), _p(r_kinv), C.c_int(src_w), C.c_int(src_h), _p(roi), _p(mm))
    return roi, mm


def detect_roi_raw(kind, scale, k, rinv, r_kinv, src_w, src_h):
    roi = np.zeros(4, np.int32)
    mm = np.zeros(4, np.float32)
    k, rinv, r_kinv = _c(k, np.float32), _c(rinv, np.float32), _c(r_kinv, np.float32)
    lib().orc_detect_roi(C.c_int(kind), C.c_float(scale), _p(k), _p(rinv), _p(r_kinv), C.c_int(src_w), C.c_int(src_h), _p(roi), _p(mm))
    return roi, mm


def build_maps(kind, scale, k_rinv, roi):
    roi = _c(roi, np.int32)
    mh, mw = int(roi[3] - roi[1] + 1), int(roi[2] - roi[0] + 1)
    xm = np.empty((mh, mw), np.float32)
    ym = np.empty((mh, mw), np.float32)
    kr = _c(k_rinv, np.float32)
    lib().orc_build_maps(C.c_int(kind), C.c_float(scale), _p(kr), _p(roi), _p(xm), _p(ym))
    return xm, ym


def remap(src, xmap, ymap, interp, border):
    src = np.ascontiguousarray(src)
    xm, ym = _c(xmap, np.float32), _c(ymap, np.float32)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dh, dw = xm.shape
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), src.dtype)
    fn = lib().orc_remap_u8 if src.dtype == np.uint8 else lib().orc_remap_f32
    assert src.dtype in (np.uint8, np.float32)
    fn(_p(src), C.c_int(sh), C.c_int(sw), C.c_int(cn), C.c_size_t(src.strides[0]),
       _p(dst), C.c_int(dh), C.c_int(dw), C.c_size_t(dst.strides[0]), _p(xm), _p(ym), C.c_int(interp), C.c_int(border))
    return dst


def warp_u8(kind, scale, K, R, src, interp, border):
    """W:145-161 on a u8 image; returns (corner(x,y), dst, roi)."""
    src = np.ascontiguousarray(src, np.uint8)
    K, R = _c(K, np.float32).reshape(9), _c(R, np.float32).reshape(9)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    roi = np.zeros(4, np.int32)
    lib().orc_warp_u8(C.c_int(kind), C.c_float(scale), _p(K), _p(R), _p(src), C.c_int(sh), C.c_int(sw), C.c_int(cn),
                      C.c_int(interp), C.c_int(border), _p(roi), None)
    dh, dw = int(roi[3] - roi[1] + 1), int(roi[2] - roi[0] + 1)
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_warp_u8(C.c_int(kind), C.c_float(scale), _p(K), _p(R), _p(src), C.c_int(sh), C.c_int(sw), C.c_int(cn),
                      C.c_int(interp), C.c_int(border), _p(roi), _p(dst))
    return (int(roi[0]), int(roi[1])), dst, roi


# ---------------------------------------------------------------- pyramids
def pyr_down(a):
    a = np.ascontiguousarray(a)
    sh, sw = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    shape = ((sh + 1) // 2, (sw + 1) // 2) + (() if a.ndim == 2 else (cn,))
    d = np.empty(shape, a.dtype)
    fn = {np.dtype(np.int16): lib().orc_pyr_down_s16, np.dtype(np.float32): lib().orc_pyr_down_f32}[a.dtype]
    fn(_p(a), C.c_int(sh), C.c_int(sw), C.c_int(cn), _p(d))
    return d


def pyr_up(a):
    a = np.ascontiguousarray(a)
    sh, sw = a.shape[:2]
    cn = 1 if a.ndim == 2 else a.shape[2]
    shape = (sh * 2, sw * 2) + (() if a.ndim == 2 else (cn,))
    d = np.empty(shape, a.dtype)
    fn = {np.dtype(np.int16): lib().orc_pyr_up_s16, np.dtype(np.float32): lib().orc_pyr_up_f32}[a.dtype]
    fn(_p(a), C.c_int(sh), C.c_int(sw), C.c_int(cn), _p(d))
    return d


def f16_round(a):
    a = _c(a, np.float32)
    out = np.empty_like(a)
    f = lib().orc_f16_round
    flat_in, flat_out = a.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = f(C.c_float(float(flat_in[i])))
    return out


# ---------------------------------------------------------------- multi-band blender
class MultiBand:
    """orc_mb_* : OpenCV 3.4.2 MultiBandBlender restated (A9, A11, A12)."""

    def __init__(self, num_bands=5, precision=I16):
        self.h = lib().orc_mb_create(num_bands, precision)
        self.precision = precision

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mb_destroy(self.h)
            self.h = None

    def prepare(self, corners, sizes):
        c = _c(np.asarray(corners).reshape(-1), np.int32)
        s = _c(np.asarray(sizes).reshape(-1), np.int32)
        lib().orc_mb_prepare(self.h, len(c) // 2, _p(c), _p(s))

    @property
    def num_bands(self):
        return lib().orc_mb_num_bands(self.h)

    def result_size(self):
        w, h = C.c_int(), C.c_int()
        lib().orc_mb_result_size(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def feed(self, img, mask, tl):
        img = np.ascontiguousarray(img)
        mask = _c(mask, np.uint8)
        assert img.dtype in (np.int16, np.float32) and img.shape[2] == 3
        lib().orc_mb_feed(self.h, _p(img), int(img.dtype == np.float32), _p(mask), img.shape[0], img.shape[1], int(tl[0]), int(tl[1]))

    def level(self, i):
        r, c = C.c_int(), C.c_int()
        lib().orc_mb_level(self.h, i, None, None, C.byref(r), C.byref(c))
        lap = np.empty((r.value, c.value, 3), np.int16 if self.precision == I16 else np.float32)
        w = np.empty((r.value, c.value), np.float32)
        lib().orc_mb_level(self.h, i, _p(lap), _p(w), C.byref(r), C.byref(c))
        return lap, w

    def blend(self, out_f32=False):
        w, h = self.result_size()
        dst = np.empty((h, w, 3), np.float32 if out_f32 else np.int16)
        m = np.empty((h, w), np.uint8)
        lib().orc_mb_blend(self.h, _p(dst), int(out_f32), _p(m))
        return dst, m


# ---------------------------------------------------------------- N3 / N2
def dilate_rect(mask, kw, kh):
    m = _c(mask, np.uint8)
    out = np.empty_like(m)
    lib().orc_dilate_rect_u8(_p(m), m.shape[0], m.shape[1], int(kw), int(kh), _p(out))
    return out


def gain_apply(img, gain):
    a = np.ascontiguousarray(img, dtype=np.uint8).copy()
    lib().orc_gain_apply_u8(_p(a), C.c_size_t(a.size), C.c_double(gain))
    return a


def distance_transform_l1(mask):
    m = _c(mask, np.uint8)
    out = np.empty(m.shape, np.float32)
    lib().orc_distance_transform_l1(_p(m), m.shape[0], m.shape[1], _p(out))
    return out


def feather_weight_map(mask, sharpness):
    m = _c(mask, np.uint8)
    out = np.empty(m.shape, np.float32)
    lib().orc_feather_weight_map(_p(m), m.shape[0], m.shape[1], C.c_float(sharpness), _p(out))
    return out


class Feather:
    """orc_fb_*: OpenCV 3.4.2 FeatherBlender restated (the blender every reference demo runs, W:278-280)."""

    def __init__(self, sharpness=0.02):
        lib().orc_fb_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_fb_create(C.c_float(sharpness)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fb_destroy(self.h)
            self.h = None

    def prepare(self, corners, sizes):
        c = _c(np.asarray(corners).reshape(-1), np.int32)
        s = _c(np.asarray(sizes).reshape(-1), np.int32)
        lib().orc_fb_prepare(self.h, len(c) // 2, _p(c), _p(s))

    def result_size(self):
        w, h = C.c_int(), C.c_int()
        lib().orc_fb_result_size(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def feed(self, img, mask, tl):
        img, mask = _c(img, np.int16), _c(mask, np.uint8)
        lib().orc_fb_feed(self.h, _p(img), _p(mask), img.shape[0], img.shape[1], int(tl[0]), int(tl[1]))

    def blend(self):
        w, h = self.result_size()
        dst, m = np.empty((h, w, 3), np.int16), np.empty((h, w), np.uint8)
        lib().orc_fb_blend(self.h, _p(dst), _p(m))
        return dst, m


class NoBlend:
    """orc_nb_*: cv::detail::Blender itself (Blender::NO, W:276) restated."""

    def __init__(self):
        lib().orc_nb_create.restype = C.c_void_p
        self.h = C.c_void_p(lib().orc_nb_create())

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_nb_destroy(self.h)
            self.h = None

    def prepare(self, corners, sizes):
        c = _c(np.asarray(corners).reshape(-1), np.int32)
        s = _c(np.asarray(sizes).reshape(-1), np.int32)
        lib().orc_nb_prepare(self.h, len(c) // 2, _p(c), _p(s))

    def result_size(self):
        w, h = C.c_int(), C.c_int()
        lib().orc_nb_result_size(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def feed(self, img, mask, tl):
        img, mask = _c(img, np.int16), _c(mask, np.uint8)
        lib().orc_nb_feed(self.h, _p(img), _p(mask), img.shape[0], img.shape[1], int(tl[0]), int(tl[1]))

    def blend(self):
        w, h = self.result_size()
        dst, m = np.empty((h, w, 3), np.int16), np.empty((h, w), np.uint8)
        lib().orc_nb_blend(self.h, _p(dst), _p(m))
        return dst, m


def convert_f32(src, dtype):
    """src.convertTo(dst, CV_16S / CV_8U) of a float image: saturate_cast = clamp(cvRound(v))  (W:294, W:315's input)."""
    src = _c(src, np.float32)
    dst = np.empty(src.shape, np.dtype(dtype))
    fn = lib().orc_convert_f32_s16 if dst.dtype == np.int16 else lib().orc_convert_f32_u8
    fn(_p(src), C.c_size_t(src.size), _p(dst))
    return dst


# ---------------------------------------------------------------- A13
def blend_pair_linear(img1, img2, tl1, tl2):
    img1, img2 = _c(img1, np.float32), _c(img2, np.float32)
    pr, pc = C.c_int(), C.c_int()
    lib().orc_blend_pair_linear_size(img1.shape[0], img1.shape[1], img2.shape[0], img2.shape[1],
                                     int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]), C.byref(pr), C.byref(pc))
    pano = np.empty((pr.value, pc.value, 3), np.float32)
    seam = np.zeros(pr.value, np.int32)
    rc = lib().orc_blend_pair_linear(_p(img1), img1.shape[0], img1.shape[1], _p(img2), img2.shape[0], img2.shape[1],
                                     int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]), _p(pano), _p(seam))
    return rc, pano, seam


def pair_linear_costv(img1, img2, tl1, tl2):
    """costV of B:207-261 (what B:265 writes to costV.bmp) -> float32 (interSectHe_, interSectBr_ + 2)."""
    img1, img2 = _c(img1, np.float32), _c(img2, np.float32)
    pr, pc = C.c_int(), C.c_int()
    lib().orc_blend_pair_linear_size(img1.shape[0], img1.shape[1], img2.shape[0], img2.shape[1],
                                     int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]), C.byref(pr), C.byref(pc))
    ibr = img1.shape[1] - (int(tl2[0]) - int(tl1[0]))
    pano = np.empty((pr.value, pc.value, 3), np.float32)
    cost = np.zeros((pr.value, ibr + 2), np.float32)
    rc = lib().orc_pair_linear_costv(_p(img1), img1.shape[0], img1.shape[1], _p(img2), img2.shape[0], img2.shape[1],
                                     int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]), _p(pano), _p(cost))
    return rc, cost


# ---------------------------------------------------------------- verbatim reference (W:30-63)
def ref_set(scale, r_kinv, k_rinv):
    r, k = _c(r_kinv, np.float32), _c(k_rinv, np.float32)
    ref().ref_set(C.c_float(scale), _p(r), _p(k))


def ref_map_forward_n(x, y):
    x, y = _c(x, np.float32), _c(y, np.float32)
    u, v = np.empty_like(x), np.empty_like(x)
    ref().ref_map_forward_n(C.c_int(x.size), _p(x), _p(y), _p(u), _p(v))
    return u, v


def ref_map_backward_n(u, v):
    u, v = _c(u, np.float32), _c(v, np.float32)
    x, y = np.empty_like(u), np.empty_like(u)
    ref().ref_map_backward_n(C.c_int(u.size), _p(u), _p(v), _p(x), _p(y))
    return x, y


def _seam_args(img1, img2, tl1, tl2, union_tl, labels, label, roi):
    is_u8 = img1.dtype == np.uint8
    a = _c(img1, np.uint8 if is_u8 else np.float32)
    b = _c(img2, np.uint8 if is_u8 else np.float32)
    lab = _c(labels, np.int32)
    return a, b, lab, [_p(a), a.shape[0], a.shape[1], _p(b), b.shape[0], b.shape[1], int(is_u8), int(tl1[0]), int(tl1[1]), int(tl2[0]), int(tl2[1]),
                       int(union_tl[0]), int(union_tl[1]), _p(lab), lab.shape[0], lab.shape[1], int(label), int(roi[0]), int(roi[1]), int(roi[2]), int(roi[3])]


def seam_costs(img1, img2, tl1, tl2, union_tl, labels, label, roi):
    """computeCosts S:733-803 -> (costV rh x (rw+1), costH (rh+1) x rw)."""
    a, b, lab, args = _seam_args(img1, img2, tl1, tl2, union_tl, labels, label, roi)
    rw, rh = int(roi[2]), int(roi[3])
    cv, ch = np.empty((rh, rw + 1), np.float32), np.empty((rh + 1, rw), np.float32)
    lib().orc_seam_costs(*args, _p(cv), _p(ch))
    return cv, ch


def seam_estimate(img1, img2, tl1, tl2, union_tl, labels, label, roi, p1, p2):
    """estimateSeam S:806-957 -> (seam (N, 2) int32, p1 first; empty = unreachable, is_horizontal)."""
    a, b, lab, args = _seam_args(img1, img2, tl1, tl2, union_tl, labels, label, roi)
    cap = int(roi[2]) + int(roi[3]) + 2
    out = np.zeros((cap, 2), np.int32)
    horiz = C.c_int(0)
    lib().orc_seam_estimate.restype = C.c_int
    n = lib().orc_seam_estimate(*args, int(p1[0]), int(p1[1]), int(p2[0]), int(p2[1]), _p(out), cap, C.byref(horiz))
    return out[:n].copy(), bool(horiz.value)
