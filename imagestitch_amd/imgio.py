"""cv::imread / cv::imwrite for the .bmp files either side of the hot path (W:166, W:155-156, W:315) — SURVEY §8(f) N4.
Reading: uncompressed Windows bitmaps (the reference's inputs and committed artefacts).  Writing: .bmp and baseline
JFIF .jpg (imwrite("pano.jpg", result), S:1282); .jpg files are read too (baseline / sequential Huffman JPEG, libjpeg's arithmetic)."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import as_mat, check


def imread(path, device=None):
    """cv::imread(path) (IMREAD_COLOR): HxWx3 uint8 BGR — a numpy array, or a torch CUDA tensor when `device` is given."""
    lib = _lib.load()
    rows, cols = C.c_int(), C.c_int()
    try:
        with open(path, "rb") as f:
            magic = f.read(2)
    except OSError:
        magic = b""                    # the library reports the unreadable file (IsxError, as for every other failure)
    jpeg = magic == b"\xff\xd8"        # like cv::imread: the decoder follows the file's signature, not its extension
    size, read = (lib.isx_jpeg_size, lib.isx_jpeg_read) if jpeg else (lib.isx_bmp_size, lib.isx_bmp_read)
    check(size(os.fsencode(path), C.byref(rows), C.byref(cols)))
    if device is None:
        out = np.empty((rows.value, cols.value, 3), np.uint8)
    else:
        import torch
        out = torch.empty((rows.value, cols.value, 3), dtype=torch.uint8, device=torch.device("cuda", device))
    m = as_mat(out)
    check(read(os.fsencode(path), C.byref(m)))
    return out


def imwrite(path, img, quality=95):
    """cv::imwrite(path, img) for CV_8UC3 / CV_8UC1 host or device mats: .bmp, or .jpg / .jpeg (baseline JFIF; `quality` is
    cv::IMWRITE_JPEG_QUALITY, OpenCV's default 95) - the format follows the extension, as in OpenCV."""
    m = as_mat(img)
    ext = os.path.splitext(os.fspath(path))[1].lower()
    if ext in (".jpg", ".jpeg", ".jpe"):
        check(_lib.load().isx_jpeg_write(os.fsencode(path), C.byref(m), int(quality)))
    else:
        check(_lib.load().isx_bmp_write(os.fsencode(path), C.byref(m)))
    return True
