"""Host-side mirror of cv::detail::RotationWarper as the reference uses it (W:217-233, B:99-110).

    warper = CylindricalWarper().create(scale)        # warper_creator->create(focal)   W:217-222
    corner, warped = warper.warp(img, K, R, INTER_LINEAR, BORDER_REFLECT)       # W:229
    _, mask_w      = warper.warp(mask, K, R, INTER_NEAREST, BORDER_CONSTANT)    # W:232

All pixel work happens in the HIP library behind the C-ABI (imagestitch_amd/csrc); this module only
marshals arguments.  numpy arrays are host mats (staged over PCIe), torch CUDA tensors are device
mats (zero-copy).  Outputs have the kind of the input.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BORDER_CONSTANT, BORDER_REFLECT, INTER_LINEAR, INTER_NEAREST, WARP_CYLINDRICAL,  # noqa: F401
                   WARP_SPHERICAL, IsxError, as_mat, check, f9)


def _is_tensor(a):
    try:
        import torch
        return isinstance(a, torch.Tensor)
    except ImportError:  # pragma: no cover
        return False


def _empty_like_kind(src, shape, dtype):
    if _is_tensor(src):
        import torch
        return torch.empty(shape, dtype=getattr(torch, np.dtype(dtype).name), device=src.device)
    return np.empty(shape, dtype)


class RotationWarper:
    """cv::detail::RotationWarper: warp / buildMaps / warpRoi on one projector kind."""

    def __init__(self, kind, scale, device=0, stream=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.isx_warper_create(kind, float(scale), int(device), C.byref(self._h)))
        self.kind, self.scale, self.device = kind, float(scale), device
        if stream is not None:
            self.set_stream(stream)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.isx_warper_destroy(h)
            self._h = None

    def set_stream(self, stream):
        """stream: a torch.cuda.Stream, an int hipStream_t, or None for the default stream."""
        ptr = getattr(stream, "cuda_stream", stream)
        check(self._lib.isx_warper_set_stream(self._h, C.c_void_p(ptr or 0)))

    def camera(self, K, R):
        """setCameraParams (W:90-120) -> (r_kinv, k_rinv) as the library computes them."""
        _k, kp = f9(K)
        _r, rp = f9(R)
        a, b = np.zeros(9, np.float32), np.zeros(9, np.float32)
        check(self._lib.isx_warper_camera(self._h, kp, rp, a.ctypes.data_as(_lib._F9), b.ctypes.data_as(_lib._F9)))
        return a, b

    def warpRoi(self, src_size, K, R, with_minmax=False):
        """detectResultRoi (W:64-88).  src_size = (width, height).  Returns (tl.x, tl.y, br.x, br.y)."""
        _k, kp = f9(K)
        _r, rp = f9(R)
        roi = (C.c_int * 4)()
        mm = (C.c_float * 4)()
        check(self._lib.isx_warper_roi(self._h, int(src_size[0]), int(src_size[1]), kp, rp, roi, mm))
        if with_minmax:
            return tuple(roi), np.array(list(mm), np.float32)
        return tuple(roi)

    def buildMapsRoi(self, K, R, roi, like=None):
        """The map fill of buildMaps (W:133-141) over a given rectangle (tl.x, tl.y, br.x, br.y) -> (xmap, ymap)."""
        shape = (roi[3] - roi[1] + 1, roi[2] - roi[0] + 1)
        xm = _empty_like_kind(like, shape, np.float32) if like is not None else np.empty(shape, np.float32)
        ym = _empty_like_kind(like, shape, np.float32) if like is not None else np.empty(shape, np.float32)
        _k, kp = f9(K)
        _r, rp = f9(R)
        mx, my = as_mat(xm), as_mat(ym)
        check(self._lib.isx_warper_build_maps_roi(self._h, kp, rp, (C.c_int * 4)(*[int(v) for v in roi]), C.byref(mx), C.byref(my)))
        return xm, ym

    def buildMaps(self, src_size, K, R, like=None):
        """buildMaps (W:122-144) -> (roi, xmap, ymap)."""
        roi = self.warpRoi(src_size, K, R)
        shape = (roi[3] - roi[1] + 1, roi[2] - roi[0] + 1)
        xm = _empty_like_kind(like, shape, np.float32) if like is not None else np.empty(shape, np.float32)
        ym = _empty_like_kind(like, shape, np.float32) if like is not None else np.empty(shape, np.float32)
        _k, kp = f9(K)
        _r, rp = f9(R)
        mx, my = as_mat(xm), as_mat(ym)
        r2 = (C.c_int * 4)()
        check(self._lib.isx_warper_build_maps(self._h, int(src_size[0]), int(src_size[1]), kp, rp, C.byref(mx), C.byref(my), r2))
        return tuple(r2), xm, ym

    def warp(self, src, K, R, interp_mode, border_mode, dst=None):
        """Point warp(src, K, R, interp, border, dst) (W:145-161) -> (corner, dst)."""
        ms = as_mat(src)
        _k, kp = f9(K)
        _r, rp = f9(R)
        if dst is None:   # one detectResultRoi per warp() (W:126): isx_warper_roi sizes dst, isx_warper_warp_roi runs the remap
            roi = self.warpRoi((ms.cols, ms.rows), K, R)
            shape = (roi[3] - roi[1] + 1, roi[2] - roi[0] + 1) + tuple(src.shape[2:])
            dst = _empty_like_kind(src, shape, np.dtype(str(src.dtype).replace("torch.", "")))
            md = as_mat(dst)
            check(self._lib.isx_warper_warp_roi(self._h, C.byref(ms), kp, rp, int(interp_mode), int(border_mode), (C.c_int * 4)(*roi), C.byref(md)))
            return (roi[0], roi[1]), dst
        md = as_mat(dst)
        corner = (C.c_int * 2)()
        check(self._lib.isx_warper_warp(self._h, C.byref(ms), kp, rp, int(interp_mode), int(border_mode), C.byref(md), corner))
        return (corner[0], corner[1]), dst

    def warp_roi(self, src, K, R, interp_mode, border_mode, roi, dst):
        """The second half of a RotationWarper::warp as include/imagestitch_cv.hpp issues it: isx_warper_warp_roi with the ROI that
        isx_warper_roi (warpRoi) just returned, into the caller's dst of (roi height + 1) x (roi width + 1) (W:150, W:157)."""
        ms, md = as_mat(src), as_mat(dst)
        _k, kp = f9(K)
        _r, rp = f9(R)
        check(self._lib.isx_warper_warp_roi(self._h, C.byref(ms), kp, rp, int(interp_mode), int(border_mode), (C.c_int * 4)(*[int(v) for v in roi]), C.byref(md)))
        return dst

    def warp_with_mask(self, img, K, R, mask=None, out16=False, dst_img=None, dst_mask=None):
        """W:229 + W:232 (+ W:294 when out16) in one pass -> (corner, warped_img, warped_mask)."""
        mi = as_mat(img)
        mm = as_mat(mask) if mask is not None else None
        _k, kp = f9(K)
        _r, rp = f9(R)
        if dst_img is None or dst_mask is None:   # one detectResultRoi for the call, as in warp()
            roi = self.warpRoi((mi.cols, mi.rows), K, R)
            h, w = roi[3] - roi[1] + 1, roi[2] - roi[0] + 1
            dst_img = _empty_like_kind(img, (h, w, 3), np.int16 if out16 else np.uint8)
            dst_mask = _empty_like_kind(img, (h, w), np.uint8)
            mdi, mdm = as_mat(dst_img), as_mat(dst_mask)
            check(self._lib.isx_warper_warp_with_mask_roi(self._h, C.byref(mi), C.byref(mm) if mm is not None else None, kp, rp,
                                                          (C.c_int * 4)(*roi), C.byref(mdi), C.byref(mdm)))
            return (roi[0], roi[1]), dst_img, dst_mask
        mdi, mdm = as_mat(dst_img), as_mat(dst_mask)
        corner = (C.c_int * 2)()
        check(self._lib.isx_warper_warp_with_mask(self._h, C.byref(mi), C.byref(mm) if mm is not None else None, kp, rp,
                                                  C.byref(mdi), C.byref(mdm), corner))
        return (corner[0], corner[1]), dst_img, dst_mask

    def warp_with_mask_planned(self, img, K, R, planned_roi, dst_img, dst_mask, mask=None):
        """Sync-free variant: caller planned the ROI (warpRoi) and allocated the outputs."""
        mi, mdi, mdm = as_mat(img), as_mat(dst_img), as_mat(dst_mask)
        mm = as_mat(mask) if mask is not None else None
        _k, kp = f9(K)
        _r, rp = f9(R)
        roi = (C.c_int * 4)(*[int(v) for v in planned_roi])
        check(self._lib.isx_warper_warp_with_mask_planned(self._h, C.byref(mi), C.byref(mm) if mm is not None else None, kp, rp,
                                                          roi, C.byref(mdi), C.byref(mdm)))

    def begin_batch(self):
        """isx_warper_begin_batch: the fused tile warps that follow are collected; end_batch() launches them as ONE kernel (blockIdx.z = tile)."""
        check(self._lib.isx_warper_begin_batch(self._h))

    def end_batch(self):
        check(self._lib.isx_warper_end_batch(self._h))

    def set_roi_cache(self, on=True):
        """Opt-in: remember detectResultRoi per (K, R, scale, source size); repeated calls skip the scan and its host sync."""
        check(self._lib.isx_warper_set_roi_cache(self._h, int(bool(on))))

    def set_gain(self, gain=1.0):
        """GainCompensator::apply (W:241-244) folded into the fused warps that follow (warp_with_mask*, all-255 mask): isx_warper_set_gain."""
        check(self._lib.isx_warper_set_gain(self._h, C.c_double(float(gain))))

    def set_dst_columns(self, col0=0, col1=0):
        """The fused warps that follow produce only the columns [col0, col1) of the warped tile (isx_warper_set_dst_columns)."""
        check(self._lib.isx_warper_set_dst_columns(self._h, int(col0), int(col1)))

    def set_deferred_verify(self, on=True):
        check(self._lib.isx_warper_set_deferred_verify(self._h, int(bool(on))))

    def verify_is_light(self, src_size, K, R):
        """isx_warper_verify_is_light: the planned warp's ROI verification is a one-workgroup border scan (no need to place it)."""
        _k, kp = f9(K)
        _r, rp = f9(R)
        v = C.c_int()
        check(self._lib.isx_warper_verify_is_light(self._h, int(src_size[0]), int(src_size[1]), kp, rp, C.byref(v)))
        return bool(v.value)

    def discard_pending(self):
        """Drop the verification scans the planned warps since the last verify() queued (a captured step: they are re-queued per replay)."""
        check(self._lib.isx_warper_discard_pending(self._h))

    def queue_verify(self, src_size, K, R, planned_roi):
        """Queue the verification of a planned ROI from the rig alone (isx_warper_queue_verify); verify() starts what is queued."""
        _k, kp = f9(K)
        _r, rp = f9(R)
        check(self._lib.isx_warper_queue_verify(self._h, int(src_size[0]), int(src_size[1]), kp, rp, (C.c_int * 4)(*[int(v) for v in planned_roi])))

    def verify(self):
        """Enqueue the queued verification scans of planned warps behind the stream's current position."""
        check(self._lib.isx_warper_verify(self._h))

    def verify_after(self, event):
        """verify(), but the scans start when `event` (torch.cuda.Event or raw hipEvent_t, already recorded) completes."""
        check(self._lib.isx_warper_verify_after(self._h, C.c_void_p(getattr(event, "cuda_event", event))))

    def join(self):
        """Make the handle's stream wait for the side-stream verification scans (needed inside graph capture)."""
        check(self._lib.isx_warper_join(self._h))

    def plan_status(self):
        n = C.c_int()
        check(self._lib.isx_warper_plan_status(self._h, C.byref(n)))
        return n.value


class _Creator:
    kind = WARP_CYLINDRICAL

    def __init__(self, device=0, stream=None):
        self.device, self.stream = device, stream

    def create(self, scale):
        return RotationWarper(self.kind, scale, self.device, self.stream)


class CylindricalWarper(_Creator):
    """cv::CylindricalWarper (W:219): warper_creator->create(scale)."""
    kind = WARP_CYLINDRICAL


class SphericalWarper(_Creator):
    """cv::SphericalWarper (B:93, commented out in the reference)."""
    kind = WARP_SPHERICAL


def remap(src, xmap, ymap, interp_mode=INTER_LINEAR, border_mode=BORDER_REFLECT, device=0, stream=None):
    """cv::remap(src, dst, xmap, ymap, interp_mode, border_mode) (W:157) with CV_32FC1 maps -> dst (same kind as src)."""
    shape = tuple(xmap.shape[:2]) + (tuple(src.shape[2:]) if len(src.shape) == 3 else ())
    dst = _empty_like_kind(src, shape, str(src.dtype).replace("torch.", "") if _is_tensor(src) else src.dtype)
    ms, mx, my, md = as_mat(src), as_mat(xmap), as_mat(ymap), as_mat(dst)
    ptr = getattr(stream, "cuda_stream", stream)
    check(_lib.load().isx_remap(C.byref(ms), C.byref(mx), C.byref(my), int(interp_mode), int(border_mode), C.byref(md), int(device), C.c_void_p(ptr or 0)))
    return dst
