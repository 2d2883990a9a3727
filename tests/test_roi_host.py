"""detectResultRoi (W:64-88) on the caller's thread (csrc/roihost.cpp, round 6): where the extrema provably lie on the source's border
`isx_warper_roi` ranks the 2 (W + H) border pixels on the host (AVX2) instead of launching a workgroup and waiting for it.  No device is
needed for that path, so the CPU suite compares it - ROI and float extrema - with the oracle's scan of EVERY source pixel (cylindrical,
W:72-81) / OpenCV's border form (spherical), for the vectorised (AVX-512F, AVX2) and the scalar code."""
import ctypes as C

import numpy as np
import pytest

from imagestitch_amd import _lib, synth

CYL, SPH = 0, 1


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def _host(lib, kind, scale, K, R, w, h, isa):
    K = np.ascontiguousarray(K, np.float32).reshape(9)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    roi = np.zeros(4, np.int32)
    mm = np.zeros(4, np.float32)
    fp = C.POINTER(C.c_float)
    rc = lib.isx_selftest_roi_host(kind, C.c_float(scale), K.ctypes.data_as(fp), R.ctypes.data_as(fp), w, h, isa,
                                   roi.ctypes.data_as(C.POINTER(C.c_int)), mm.ctypes.data_as(fp))
    return rc, roi, mm


def _rot(yaw, pitch, roll):
    return (synth._rot("y", yaw) @ synth._rot("x", pitch) @ synth._rot("z", roll)).astype(np.float32)


def _check(lib, oracle, kind, scale, K, R, w, h):
    oroi, omm = oracle.detect_roi(kind, scale, K, R, w, h)
    for isa in (0, 1, 2):       # the best vector form the CPU has (AVX-512F / AVX2), the scalar form, the AVX2 form
        rc, roi, mm = _host(lib, kind, scale, K, R, w, h, isa)
        assert rc == 0, lib.isx_last_error()
        assert np.array_equal(roi, oroi), (isa, roi, oroi)
        assert np.array_equal(mm, omm), (isa, mm, omm)


def test_config2_cameras_at_full_size(lib, oracle):
    W, H, F = 3840, 2160, 3000.0
    K, Rs = synth.camera_pair(W, H, F)
    for R in Rs:
        _check(lib, oracle, CYL, F, K, R, W, H)


def test_the_reference_rig(lib, oracle):
    # f = 2707.47 (W:30), principal point 550.5, R = I on a 1101 x 1101 source: ROI width 1086
    K = np.array([[2707.47, 0, 550.5], [0, 2707.47, 550.5], [0, 0, 1]], np.float32)
    _check(lib, oracle, CYL, 2707.47, K, np.eye(3, dtype=np.float32), 1101, 1101)
    rc, roi, _ = _host(lib, CYL, 2707.47, K, np.eye(3, dtype=np.float32), 1101, 1101, 0)
    assert rc == 0 and roi[2] - roi[0] + 1 == 1086


def test_random_cylindrical_cameras(lib, oracle):
    rng = np.random.default_rng(20261006)
    done = refused = 0
    for i in range(80):
        w, h = int(rng.integers(33, 700)), int(rng.integers(33, 500))
        f = float(rng.uniform(0.4, 3.0) * max(w, h))
        K = np.array([[f, 0, w / 2 + rng.uniform(-20, 20)], [0, f * rng.uniform(0.9, 1.1), h / 2 + rng.uniform(-20, 20)], [0, 0, 1]], np.float32)
        R = _rot(rng.uniform(-0.9, 0.9), rng.uniform(-0.5, 0.5), rng.uniform(-0.4, 0.4))
        scale = float(f * rng.uniform(0.5, 2.0))
        rc, _, _ = _host(lib, CYL, scale, K, R, w, h, 0)
        if rc != 0:
            refused += 1      # outside the proof: isx_warper_roi scans every pixel on the device there
            continue
        _check(lib, oracle, CYL, scale, K, R, w, h)
        done += 1
    assert done >= 40, (done, refused)


def test_a_camera_that_looks_along_the_axis_is_refused(lib):
    K = np.array([[300, 0, 160], [0, 300, 120], [0, 0, 1]], np.float32)
    rc, _, _ = _host(lib, CYL, 300.0, K, _rot(0.0, 1.45, 0.0), 320, 240, 0)     # the cylinder's pole inside the image
    assert rc != 0
    rc, _, _ = _host(lib, CYL, 300.0, K, _rot(2.0, 0.0, 0.0), 320, 240, 0)      # part of the image behind the camera
    assert rc != 0


def test_spherical_cameras_incl_poles(lib, oracle):
    rng = np.random.default_rng(7)
    W, H, F = 960, 540, 750.0
    K, Rs = synth.camera_ring(W, H, F, 8, 0.55)
    for R in Rs:
        _check(lib, oracle, SPH, F, K, R, W, H)
    for i in range(40):
        w, h = int(rng.integers(40, 500)), int(rng.integers(40, 400))
        f = float(rng.uniform(0.5, 2.0) * max(w, h))
        K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
        R = _rot(rng.uniform(-3.1, 3.1), rng.uniform(-1.6, 1.6), rng.uniform(-0.5, 0.5))    # now and then a pole inside the image
        _check(lib, oracle, SPH, float(f * rng.uniform(0.6, 1.5)), K, R, w, h)
