"""Host-side mirror of cv::detail::Blender / MultiBandBlender as the reference calls it
(W:271-273, 281, 302, 313; S:1244-1280).

    blender = Blender.createDefault(Blender.MULTI_BAND, False)      # W:271
    blender.setNumBands(4)                                          # W:273
    blender.prepare(corners, sizes)                                 # W:281
    blender.feed(img_s16, mask, corner)                             # W:302
    result, result_mask = blender.blend()                           # W:313
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BLEND_MULTI_BAND, PREC_F16ACC32, PREC_F32, PREC_I16, IsxError, as_mat, check  # noqa: F401
from .warper import _empty_like_kind, _is_tensor


class MultiBandBlender:
    def __init__(self, try_gpu=False, num_bands=5, precision=PREC_I16, device=0, stream=None):
        del try_gpu  # the reference passes false everywhere (W:276,278); this IS the accelerator path
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.isx_blender_create(BLEND_MULTI_BAND, int(num_bands), int(precision), int(device), C.byref(self._h)))
        self.precision = precision
        self._like = None
        self._stream_obj = None
        self._deferred_refs = False   # deferred mode 1: the C side reads the fed device mats in blend()
        self._fed = []                # ... so they are kept alive here from feed() to blend() / the next prepare()
        if stream is not None:
            self.set_stream(stream)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.isx_blender_destroy(h)
            self._h = None

    def _keep(self, img, mask):
        """Deferred mode 1 records raw device pointers of the fed mats and reads them in blend(): a temporary such as
        `torch.from_numpy(m).to(dev)` passed to feed() would otherwise be freed and its memory recycled by torch's caching
        allocator before blend() runs.  (Mode 2 copies, the eager cycle consumes at once: nothing to keep.)"""
        self._like = img
        if self._deferred_refs:
            self._fed.append((img, mask))

    def set_stream(self, stream):
        ptr = getattr(stream, "cuda_stream", stream)
        self._stream_obj = stream if hasattr(stream, "cuda_stream") else None
        check(self._lib.isx_blender_set_stream(self._h, C.c_void_p(ptr or 0)))

    def set_deferred_level0(self, on=True):
        """Opt-in (see include/imagestitch_hip.h): True / 1 = fed device mats must stay valid until blend() returns;
        "copy" / 2 = feed() takes private copies of them (OpenCV's contract: feed consumes its inputs)."""
        mode = 2 if on in ("copy", 2) else int(bool(on))
        check(self._lib.isx_blender_set_deferred_level0(self._h, mode))
        self._deferred_refs = mode == 1

    def set_mark_event(self, event, after_level=0):
        """A deferred blend() records `event` (torch.cuda.Event / hipEvent_t / None) right after its pyrDown launch of
        level `after_level`: see isx_blender_set_mark_event."""
        self._mark = event   # keep it alive
        ptr = getattr(event, "cuda_event", event) if event is not None else None
        check(self._lib.isx_blender_set_mark_event(self._h, C.c_void_p(ptr or 0), int(after_level)))

    def set_overlap(self, on=True):
        check(self._lib.isx_blender_set_overlap(self._h, int(bool(on))))

    def set_window(self, x0=0, x1=0):
        """blend() of a deferred cycle produces the result's columns [x0, x1) only (x0 a multiple of WINDOW_GRANULE), into mats
        x1 - x0 wide, bit-identical to the same columns of the whole blend: one strip of a panorama that is cut across GPUs
        (isx_blender_set_window; mosaic.strip_windows / tiles_for_window).  (0, 0) removes the window."""
        check(self._lib.isx_blender_set_window(self._h, int(x0), int(x1)))
        self._window = (int(x0), int(x1)) if x1 > x0 else None

    def setNumBands(self, n):
        check(self._lib.isx_blender_set_num_bands(self._h, int(n)))

    def numBands(self):
        n = C.c_int()
        check(self._lib.isx_blender_num_bands(self._h, C.byref(n)))
        return n.value

    def prepare(self, corners, sizes=None):
        """prepare(corners, sizes) (W:281) or prepare((x, y, w, h)) = MultiBandBlender::prepare(Rect)."""
        self._fed = []
        if sizes is None:
            x, y, w, h = [int(v) for v in corners]
            check(self._lib.isx_blender_prepare_roi(self._h, x, y, w, h))
            return
        c = np.ascontiguousarray(np.asarray(corners, np.int32).reshape(-1))
        s = np.ascontiguousarray(np.asarray(sizes, np.int32).reshape(-1))
        if c.size != s.size or c.size % 2:
            raise IsxError(1, "prepare: corners and sizes must both hold n (x, y) pairs")
        check(self._lib.isx_blender_prepare(self._h, c.size // 2, c.ctypes.data_as(_lib._IP), s.ctypes.data_as(_lib._IP)))

    def feed(self, img, mask, tl):
        """feed(img CV_16SC3 [or CV_32FC3 in the float precisions], mask CV_8U, tl) (W:302)."""
        mi, mm = as_mat(img), as_mat(mask)
        self._keep(img, mask)
        check(self._lib.isx_blender_feed(self._h, C.byref(mi), C.byref(mm), int(tl[0]), int(tl[1])))

    def feed_u8(self, img, mask, tl):
        """convertTo(CV_16S) (W:294) fused into feed: img is CV_8UC3."""
        mi, mm = as_mat(img), as_mat(mask)
        self._keep(img, mask)
        check(self._lib.isx_blender_feed_u8(self._h, C.byref(mi), C.byref(mm), int(tl[0]), int(tl[1])))

    def feed_dilated(self, img, seam_mask, warped_mask, kw, kh, tl):
        """W:286-302 in one call: feed(img, dilate(seam_mask, MORPH_RECT kw x kh) & warped_mask, tl) (isx_blender_feed_dilated)."""
        mi, ms, mw = as_mat(img), as_mat(seam_mask), as_mat(warped_mask)
        self._keep(img, None)
        check(self._lib.isx_blender_feed_dilated(self._h, C.byref(mi), C.byref(ms), C.byref(mw), int(kw), int(kh), int(tl[0]), int(tl[1])))

    def result_size(self):
        w, h = C.c_int(), C.c_int()
        check(self._lib.isx_blender_result_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def last_path(self):
        """isx_blender_last_path: which kernels the last blend() ran - {"cycle": eager | deferred | deferred_batched | deferred_strips | deferred_table, "last_step": none |
        collapse | collapse_gather | collapse_roll}."""
        c, k = C.c_int(), C.c_int()
        check(self._lib.isx_blender_last_path(self._h, C.byref(c), C.byref(k)))
        return {"cycle": ("eager", "deferred", "deferred_batched", "deferred_strips", "deferred_table")[c.value], "last_step": ("none", "collapse", "collapse_gather", "collapse_roll")[k.value]}

    def level1_format(self):
        """isx_blender_level1_format: layout of the tiles' level 1 in the last deferred blend() - records | planar | planar_q8."""
        f = C.c_int()
        check(self._lib.isx_blender_level1_format(self._h, C.byref(f)))
        return ("records", "planar", "planar_q8")[f.value]

    def table_uploads(self):
        """isx_blender_table_uploads: 3.5 KB pieces of tile tables uploaded so far (cycle deferred_table; a fixed rig uploads once)."""
        n = C.c_longlong()
        check(self._lib.isx_blender_table_uploads(self._h, C.byref(n)))
        return n.value

    def set_narrow_copies(self, on=True):
        """isx_blender_set_narrow_copies: False = private copies of CV_16SC3 tiles stay CV_16SC3 and blend() never waits for the GPU
        (with narrowed copies - the default - it polls one pinned word the first launch of its chain publishes)."""
        check(self._lib.isx_blender_set_narrow_copies(self._h, 1 if on else 0))

    def feed_path(self):
        """isx_blender_feed_path: how the last blend()'s tiles were fed in mode 2 - {"fused_tiles": n, "narrowed": none | confirmed | widened}."""
        f, n = C.c_int(), C.c_int()
        check(self._lib.isx_blender_feed_path(self._h, C.byref(f), C.byref(n)))
        return {"fused_tiles": f.value, "narrowed": ("none", "confirmed", "widened")[n.value]}

    def level(self, i):
        """Accumulated destination pyramid level i (parity tests): (laplacian HxWx3, weight HxW)."""
        r, c = C.c_int(), C.c_int()
        check(self._lib.isx_blender_debug_level(self._h, int(i), None, None, C.byref(r), C.byref(c)))
        lap = np.empty((r.value, c.value, 3), np.int16 if self.precision == PREC_I16 else np.float32)
        w = np.empty((r.value, c.value), np.float32)
        check(self._lib.isx_blender_debug_level(self._h, int(i), lap.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p), C.byref(r), C.byref(c)))
        return lap, w

    def blend(self, dst=None, dst_mask=None, out_f32=False, out_u8=False):
        """blend(result, result_mask) (W:313) -> (result, result_mask).  out_u8: + result.convertTo(CV_8U)."""
        w, h = self.result_size()
        if getattr(self, "_window", None):
            w = self._window[1] - self._window[0]
        like = self._like if (self._like is not None and _is_tensor(self._like)) else np.empty(0)
        if dst is None:
            dst = _empty_like_kind(like, (h, w, 3), np.uint8 if out_u8 else (np.float32 if out_f32 else np.int16))
        if dst_mask is None:
            dst_mask = _empty_like_kind(like, (h, w), np.uint8)
        md, mm = as_mat(dst), as_mat(dst_mask)
        try:
            check(self._lib.isx_blender_blend(self._h, C.byref(md), C.byref(mm)))
        finally:
            # blend() only ENQUEUES the kernels that read the fed mats; torch frees a tensor's block for reuse on the stream it
            # was allocated on, which is safe as long as that is the blender's stream - record the use for other streams
            for img, mask in self._fed:
                for t in (img, mask):
                    if _is_tensor(t) and t.is_cuda and self._stream_obj is not None and hasattr(t, "record_stream"):
                        t.record_stream(self._stream_obj)
            self._fed = []
        return dst, dst_mask


def blend_batch(blenders, dsts, dst_masks):
    """isx_blender_blend_batch: blend() of several prepared + fed blenders in one chain of launches (deferred multi-band cycles of one
    rig share every launch; the rest are blended one by one).  dsts / dst_masks: one output per blender.  Returns (dsts, dst_masks)."""
    n = len(blenders)
    lib = blenders[0]._lib
    hs = (C.c_void_p * n)(*[b._h for b in blenders])
    mat_t = type(as_mat(dsts[0]))
    md = (mat_t * n)(*[as_mat(d) for d in dsts])
    mm = (mat_t * n)(*[as_mat(m) for m in dst_masks])
    try:
        check(lib.isx_blender_blend_batch(hs, n, md, mm))
    finally:
        for b in blenders:
            for img, mask in b._fed:
                for t in (img, mask):
                    if _is_tensor(t) and t.is_cuda and b._stream_obj is not None and hasattr(t, "record_stream"):
                        t.record_stream(b._stream_obj)
            b._fed = []
    return dsts, dst_masks


class FeatherBlender(MultiBandBlender):
    """cv::detail::FeatherBlender as every reference demo runs it (W:278-281,302,313):
    blender = Blender.createDefault(Blender.FEATHER, False); blender.setSharpness(0.1)."""

    def __init__(self, try_gpu=False, sharpness=0.02, device=0, stream=None):
        del try_gpu
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.isx_blender_create(_lib.BLEND_FEATHER, 0, PREC_I16, int(device), C.byref(self._h)))
        self.precision = PREC_I16
        self._like = None
        self._stream_obj = None
        self._deferred_refs = False
        self._fed = []
        if stream is not None:
            self.set_stream(stream)
        self.setSharpness(sharpness)

    def setSharpness(self, val):
        check(self._lib.isx_blender_set_sharpness(self._h, float(val)))


def dilate_and(mask, kw, kh, other=None, device=0, stream=None):
    """dilate(mask, getStructuringElement(MORPH_RECT, Size(kw, kh))) [& other]  (W:286-301)."""
    out = _empty_like_kind(mask, tuple(mask.shape), np.uint8)
    mm, mo = as_mat(mask), as_mat(out)
    mt = as_mat(other) if other is not None else None
    ptr = getattr(stream, "cuda_stream", stream)
    check(_lib.load().isx_mask_dilate_and(C.byref(mm), C.byref(mt) if mt is not None else None, int(kw), int(kh), C.byref(mo), int(device), C.c_void_p(ptr or 0)))
    return out


def gain_apply(image, gain, device=0, stream=None):
    """GainCompensator::apply: multiply(image, gain, image) in place on a CV_8U image (W:241-244).  Returns image."""
    mi = as_mat(image)
    ptr = getattr(stream, "cuda_stream", stream)
    check(_lib.load().isx_gain_apply(C.byref(mi), C.c_double(float(gain)), int(device), C.c_void_p(ptr or 0)))
    return image


class NoBlender(MultiBandBlender):
    """cv::detail::Blender itself, what Blender::createDefault(Blender::NO, false) returns (W:276): feed() copies the tile's pixels
    under its mask into the CV_16SC3 accumulator, blend() zeroes what no mask covered."""

    def __init__(self, try_gpu=False, device=0, stream=None):
        del try_gpu
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.isx_blender_create(_lib.BLEND_NO, 0, PREC_I16, int(device), C.byref(self._h)))
        self.precision = PREC_I16
        self._like = None
        self._stream_obj = None
        self._deferred_refs = False
        self._fed = []
        if stream is not None:
            self.set_stream(stream)


def convert_to(src, dtype, dst=None, device=0, stream=None):
    """src.convertTo(dst, depth) with alpha = 1, beta = 0 between uint8 / int16 / float32 (W:261, W:294, W:315's input): isx_convert_to.
    dtype: numpy dtype (or its name).  Returns dst (same kind as src: numpy array or torch tensor)."""
    dt = np.dtype(dtype)
    if dst is None:
        dst = _empty_like_kind(src, tuple(src.shape), dt)
    ms, md = as_mat(src), as_mat(dst)
    ptr = getattr(stream, "cuda_stream", stream)
    check(_lib.load().isx_convert_to(C.byref(ms), C.byref(md), int(device), C.c_void_p(ptr or 0)))
    return dst


class Blender:
    """cv::detail::Blender factory (W:271,276,278)."""
    NO, FEATHER, MULTI_BAND = 0, 1, 2

    @staticmethod
    def createDefault(blend_type, try_gpu=False, **kw):
        if blend_type == Blender.FEATHER:
            return FeatherBlender(try_gpu, **kw)
        if blend_type == Blender.NO:
            return NoBlender(try_gpu, **kw)
        if blend_type != Blender.MULTI_BAND:
            raise IsxError(1, "Blender::createDefault: NO (0), FEATHER (1) or MULTI_BAND (2), got %d" % blend_type)
        return MultiBandBlender(try_gpu, **kw)
