"""ctypes loader of the C-ABI library (include/imagestitch_hip.h).

The HIP library is the product: if it is missing or fails to load this module raises — there is
no CPU fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libimagestitch_hip.so")

# ---- enums (numeric values of include/imagestitch_hip.h == OpenCV's) -------------------------
ISX_8UC1, ISX_8UC3, ISX_16SC3, ISX_32FC1, ISX_32FC3 = 0, 16, 19, 5, 21
ISX_32SC1 = 4
INTER_NEAREST, INTER_LINEAR = 0, 1
INTER_TIES_EVEN = 0x100   # OR to INTER_LINEAR: OpenCV's OpenCL (UMat) remap rounding, half to even
BORDER_CONSTANT, BORDER_REPLICATE, BORDER_REFLECT, BORDER_WRAP, BORDER_REFLECT_101 = 0, 1, 2, 3, 4
WARP_CYLINDRICAL, WARP_SPHERICAL = 0, 1
BLEND_NO, BLEND_FEATHER, BLEND_MULTI_BAND = 0, 1, 2
PREC_I16, PREC_F32, PREC_F16ACC32 = 0, 1, 2
WINDOW_GRANULE = 128       # ISX_WINDOW_GRANULE: a column window of blend() starts on a multiple of it

STATUS_NAMES = {0: "ISX_OK", 1: "ISX_ERR_INVALID", 2: "ISX_ERR_TYPE", 3: "ISX_ERR_STATE", 4: "ISX_ERR_HIP",
                5: "ISX_ERR_NOMEM", 6: "ISX_ERR_UNSUPPORTED", 7: "ISX_ERR_SIZE", 8: "ISX_ERR_PLAN", 9: "ISX_ERR_INTERNAL"}


class IsxMat(C.Structure):
    _fields_ = [("data", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("type", C.c_int),
                ("step", C.c_size_t), ("device", C.c_int)]


class IsxError(RuntimeError):
    """The cv::Exception analogue: raised for every non-zero status of the C-ABI."""

    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, code), msg))
        self.code = code
        self.msg = msg


_lib = None

_F9 = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int)
_MP = C.POINTER(IsxMat)
_SIGS = {
    "isx_device_count": [_IP],
    "isx_warper_create": [C.c_int, C.c_float, C.c_int, C.POINTER(C.c_void_p)],
    "isx_warper_destroy": [C.c_void_p],
    "isx_warper_set_stream": [C.c_void_p, C.c_void_p],
    "isx_warper_camera": [C.c_void_p, _F9, _F9, _F9, _F9],
    "isx_warper_roi": [C.c_void_p, C.c_int, C.c_int, _F9, _F9, _IP, _F9],
    "isx_warper_build_maps": [C.c_void_p, C.c_int, C.c_int, _F9, _F9, _MP, _MP, _IP],
    "isx_warper_build_maps_roi": [C.c_void_p, _F9, _F9, _IP, _MP, _MP],
    "isx_warper_warp": [C.c_void_p, _MP, _F9, _F9, C.c_int, C.c_int, _MP, _IP],
    "isx_warper_warp_roi": [C.c_void_p, _MP, _F9, _F9, C.c_int, C.c_int, _IP, _MP],
    "isx_warper_warp_with_mask_roi": [C.c_void_p, _MP, _MP, _F9, _F9, _IP, _MP, _MP],
    "isx_warper_warp_with_mask": [C.c_void_p, _MP, _MP, _F9, _F9, _MP, _MP, _IP],
    "isx_warper_warp_with_mask_planned": [C.c_void_p, _MP, _MP, _F9, _F9, _IP, _MP, _MP],
    "isx_warper_begin_batch": [C.c_void_p],
    "isx_warper_end_batch": [C.c_void_p],
    "isx_warper_plan_status": [C.c_void_p, _IP],
    "isx_warper_join": [C.c_void_p],
    "isx_warper_set_deferred_verify": [C.c_void_p, C.c_int],
    "isx_remap": [_MP, _MP, _MP, C.c_int, C.c_int, _MP, C.c_int, C.c_void_p],
    "isx_warper_set_roi_cache": [C.c_void_p, C.c_int],
    "isx_warper_set_gain": [C.c_void_p, C.c_double],
    "isx_blender_feed_dilated": [C.c_void_p, _MP, _MP, _MP, C.c_int, C.c_int, C.c_int, C.c_int],
    "isx_warper_verify": [C.c_void_p],
    "isx_warper_discard_pending": [C.c_void_p],
    "isx_warper_queue_verify": [C.c_void_p, C.c_int, C.c_int, _F9, _F9, _IP],
    "isx_warper_verify_is_light": [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)],
    "isx_warper_verify_after": [C.c_void_p, C.c_void_p],
    "isx_blender_set_mark_event": [C.c_void_p, C.c_void_p, C.c_int],
    "isx_blender_set_window": [C.c_void_p, C.c_int, C.c_int],
    "isx_warper_set_dst_columns": [C.c_void_p, C.c_int, C.c_int],
    "isx_blend_pair_linear_release": [],
    "isx_blender_create": [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "isx_blender_destroy": [C.c_void_p],
    "isx_blender_set_stream": [C.c_void_p, C.c_void_p],
    "isx_blender_set_num_bands": [C.c_void_p, C.c_int],
    "isx_blender_num_bands": [C.c_void_p, _IP],
    "isx_blender_set_deferred_level0": [C.c_void_p, C.c_int],
    "isx_blender_set_sharpness": [C.c_void_p, C.c_float],
    "isx_mask_dilate_and": [_MP, _MP, C.c_int, C.c_int, _MP, C.c_int, C.c_void_p],
    "isx_gain_apply": [_MP, C.c_double, C.c_int, C.c_void_p],
    "isx_convert_to": [_MP, _MP, C.c_int, C.c_void_p],
    "isx_seam_estimate": [_MP, _MP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _MP, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                          C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_void_p],
    "isx_dp_seam_find": [C.c_int, _MP, C.POINTER(C.c_int), _MP, C.c_int, C.c_void_p],
    "isx_dp_seam_release": [],
    "isx_bmp_size": [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "isx_bmp_read": [C.c_char_p, _MP],
    "isx_bmp_write": [C.c_char_p, _MP],
    "isx_jpeg_write": [C.c_char_p, _MP, C.c_int],
    "isx_jpeg_size": [C.c_char_p, _IP, _IP],
    "isx_jpeg_read": [C.c_char_p, _MP],
    "isx_blender_set_overlap": [C.c_void_p, C.c_int],
    "isx_blender_prepare": [C.c_void_p, C.c_int, _IP, _IP],
    "isx_blender_prepare_roi": [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int],
    "isx_blender_feed": [C.c_void_p, _MP, _MP, C.c_int, C.c_int],
    "isx_blender_feed_u8": [C.c_void_p, _MP, _MP, C.c_int, C.c_int],
    "isx_blender_result_size": [C.c_void_p, _IP, _IP],
    "isx_blender_last_path": [C.c_void_p, _IP, _IP],
    "isx_blender_level1_format": [C.c_void_p, _IP],
    "isx_blender_table_uploads": [C.c_void_p, C.POINTER(C.c_longlong)],
    "isx_blender_feed_path": [C.c_void_p, _IP, _IP],
    "isx_blender_set_narrow_copies": [C.c_void_p, C.c_int],
    "isx_blender_blend": [C.c_void_p, _MP, _MP],
    "isx_blender_blend_batch": [C.POINTER(C.c_void_p), C.c_int, _MP, _MP],
    "isx_blender_debug_level": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, _IP, _IP],
    "isx_blend_pair_linear_size": [C.c_int] * 8 + [_IP, _IP],
    "isx_blend_pair_linear": [_MP, _MP, C.c_int, C.c_int, C.c_int, C.c_int, _MP, _IP, C.c_int, C.c_void_p],
    "isx_gather_unique_id": [C.c_char_p],
    "isx_gather_create": [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)],
    "isx_gather_destroy": [C.c_void_p],
    "isx_gather_info": [C.c_void_p, _IP, _IP],
    "isx_gather_all": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p],
    "isx_gather_chunk": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p],
    "isx_gather_chunk_ptr": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)],
    "isx_gather_wait": [C.c_void_p, C.c_void_p],
    "isx_gather_synchronize": [C.c_void_p],
    "isx_gather_p2p_alloc": [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p],
    "isx_gather_p2p_open": [C.c_void_p, C.c_char_p],
    "isx_gather_p2p_chunk": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p],
    "isx_gather_p2p_wait": [C.c_void_p, C.c_void_p],
    "isx_gather_p2p_synchronize": [C.c_void_p],
    "isx_selftest_division": [C.c_int, C.c_int, C.c_ulonglong, _IP],
    "isx_selftest_roi_host": [C.c_int, C.c_float, _F9, _F9, C.c_int, C.c_int, C.c_int, _IP, _F9],
    "isx_selftest_exception_barrier": [C.c_int],
    "isx_profile_enable": [C.c_int],
    "isx_profile_reset": [],
    "isx_profile_filter": [C.c_char_p],
    "isx_profile_sample": [C.c_int],
    "isx_profile_collect": [],
    "isx_profile_count": [_IP],
    "isx_profile_entry": [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double)],
}


def declared_symbols():
    """Every entry point include/imagestitch_hip.h declares (used by the ABI export test)."""
    return sorted(list(_SIGS) + ["isx_last_error", "isx_version"])


def load():
    """Load libimagestitch_hip.so.  Import torch first in a torch process so that both share one
    libamdhip64 (same SONAME); standalone C/C++ callers get /opt/rocm/lib via the rpath."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("imagestitch_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.isx_last_error.restype = C.c_char_p
    lib.isx_last_error.argtypes = []
    lib.isx_version.restype = C.c_char_p
    lib.isx_version.argtypes = []
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise IsxError(rc, load().isx_last_error().decode("utf-8", "replace"))


def f9(a):
    import numpy as np
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(9))
    return arr, arr.ctypes.data_as(_F9)


_NP_TYPES = {("uint8", 1): ISX_8UC1, ("uint8", 3): ISX_8UC3, ("int16", 3): ISX_16SC3, ("float32", 1): ISX_32FC1, ("float32", 3): ISX_32FC3, ("int32", 1): ISX_32SC1}


def as_mat(a):
    """numpy array (host mat, device = -1) or torch CUDA tensor (device mat) -> IsxMat.
    Accepts HxW (1 channel) or HxWx{1,3}; the last dimension must be dense."""
    try:
        import torch
        is_t = isinstance(a, torch.Tensor)
    except ImportError:  # pragma: no cover
        is_t = False
    if is_t:
        cn = 1 if a.dim() == 2 else a.shape[2]
        key = (str(a.dtype).replace("torch.", ""), cn)
        if key not in _NP_TYPES:
            raise IsxError(2, "unsupported tensor dtype/channels %s" % (key,))
        es = a.element_size()
        if a.dim() == 3 and (a.stride(2) != 1 or a.stride(1) != cn):
            raise IsxError(1, "tensor must be HWC with dense pixels")
        if a.dim() == 2 and a.stride(1) != 1:
            raise IsxError(1, "tensor rows must be dense")
        dev = a.device.index if a.is_cuda else -1
        return IsxMat(a.data_ptr(), a.shape[0], a.shape[1], _NP_TYPES[key], a.stride(0) * es, dev if dev is not None else 0)
    import numpy as np
    a = np.asarray(a)
    cn = 1 if a.ndim == 2 else a.shape[2]
    key = (a.dtype.name, cn)
    if key not in _NP_TYPES:
        raise IsxError(2, "unsupported array dtype/channels %s" % (key,))
    if a.ndim == 3 and (a.strides[2] != a.itemsize or a.strides[1] != cn * a.itemsize):
        raise IsxError(1, "array must be HWC with dense pixels")
    if a.ndim == 2 and a.strides[1] != a.itemsize:
        raise IsxError(1, "array rows must be dense")
    return IsxMat(a.ctypes.data, a.shape[0], a.shape[1], _NP_TYPES[key], a.strides[0], -1)


def profile_entries():
    lib = load()
    check(lib.isx_profile_collect())
    n = C.c_int()
    check(lib.isx_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name, cnt, ms, by = C.c_char_p(), C.c_longlong(), C.c_double(), C.c_double()
        check(lib.isx_profile_entry(i, C.byref(name), C.byref(cnt), C.byref(ms), C.byref(by)))
        out[name.value.decode()] = {"launches": cnt.value, "ms": ms.value, "alg_bytes": by.value}
    return out
