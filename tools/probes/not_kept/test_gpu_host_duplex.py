"""warp() on HOST mats of a cylindrical tile, banded with upload and download overlapped (isx_warper_set_host_duplex):
same bytes as the serial path and as the oracle - upright, rolled, pitched and wide-angle cameras (the band -> last-source-row rule),
the three call forms (W:229 + W:232 fused, the image warp, the mask warp), CV_16SC3 output, a folded gain, pinned buffers."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu
CYL = 0
NEAREST, LINEAR = 0, 1
CONST, REFLECT = 0, 2


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float32)
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)


CAMS = {
    "upright": lambda R: R,
    "rolled_180": lambda R: (R @ _rot("z", np.pi)).astype(np.float32),        # the top destination rows read the LAST source rows
    "rolled_90": lambda R: (R @ _rot("z", np.pi / 2)).astype(np.float32),
    "pitched": lambda R: (_rot("x", 0.35) @ R).astype(np.float32),           # the tile bends: bands reach well past their own share of rows
    "pitched_down_rolled": lambda R: (_rot("x", -0.5) @ R @ _rot("z", 0.4)).astype(np.float32),
}


@pytest.mark.parametrize("cam", sorted(CAMS))
@pytest.mark.parametrize("focal_frac", [0.8, 0.3])
def test_banded_host_warp_equals_serial_and_oracle(gpu, oracle, cam, focal_frac):
    w, h = 1280, 720
    f = focal_frac * w
    K, Rs = synth.camera_pair(w, h, f, yaw=0.3)
    R = CAMS[cam](Rs[1])
    img = synth.make_tile(h, w, 21)
    src_mask = np.full((h, w), 255, np.uint8)
    warper = gpu.CylindricalWarper().create(f)
    roi = warper.warpRoi((w, h), K, R)
    dh, dw = roi[3] - roi[1] + 1, roi[2] - roi[0] + 1
    if dh < 512 or dh * dw > 40e6:
        pytest.skip("warped tile of %d x %d: outside the banded path's range / too large for the oracle here" % (dw, dh))
    o_corner, o_img, _ = oracle.warp_u8(CYL, f, K, R, img, LINEAR, REFLECT)
    _, o_mask, _ = oracle.warp_u8(CYL, f, K, R, src_mask, NEAREST, CONST)
    res = {}
    for on in (True, False):
        warper.set_host_duplex(on)
        c, wi, wm = warper.warp_with_mask(img, K, R)
        bands_fused = warper.last_host_bands()
        _, wi16, wm16 = warper.warp_with_mask(img, K, R, out16=True)
        li = warper.warp_roi(img, K, R, LINEAR, REFLECT, roi, np.empty((dh, dw, 3), np.uint8))
        bands_img = warper.last_host_bands()
        lm = warper.warp_roi(src_mask, K, R, NEAREST, CONST, roi, np.empty((dh, dw), np.uint8))
        bands_mask = warper.last_host_bands()
        warper.set_gain(1.37)
        _, wg, _ = warper.warp_with_mask(img, K, R)
        warper.set_gain(1.0)
        assert c == o_corner
        assert (bands_fused >= 2 and bands_img >= 2 and bands_mask >= 2) if on else (bands_fused == bands_img == bands_mask == 0)
        res[on] = (wi, wm, wi16, wm16, li, lm, wg)
    for a, b in zip(res[True], res[False]):
        assert np.array_equal(a, b)
    wi, wm, wi16, wm16, li, lm, wg = res[True]
    assert np.array_equal(wi, o_img) and np.array_equal(li, o_img) and np.array_equal(wi16, o_img.astype(np.int16))
    assert np.array_equal(wm, o_mask) and np.array_equal(lm, o_mask) and np.array_equal(wm16, o_mask)
    assert np.array_equal(wg, oracle.gain_apply(o_img, 1.37))


def test_banded_host_warp_full_size_pinned_and_pageable(gpu, oracle):
    """A 4K tile (BASELINE config 2's) through the banded path from pageable and from pinned buffers; the serial path is the reference
    (itself checked against the oracle at this size by test_gpu_blend / test_gpu_configs)."""
    import torch
    w, h, f = 3840, 2160, 3000.0
    K, Rs = synth.camera_pair(w, h, f)
    img = synth.make_tile(h, w, 3)
    warper = gpu.CylindricalWarper().create(f)
    warper.set_host_duplex(False)
    c0, wi0, wm0 = warper.warp_with_mask(img, K, Rs[0])
    assert warper.last_host_bands() == 0
    warper.set_host_duplex(True)
    c1, wi1, wm1 = warper.warp_with_mask(img, K, Rs[0])
    assert warper.last_host_bands() >= 4
    assert c0 == c1 and np.array_equal(wi0, wi1) and np.array_equal(wm0, wm1)
    pin = torch.empty((h, w, 3), dtype=torch.uint8).pin_memory()
    pin.numpy()[:] = img
    di = torch.empty(wi0.shape, dtype=torch.uint8).pin_memory()
    dm = torch.empty(wm0.shape, dtype=torch.uint8).pin_memory()
    c2, _, _ = warper.warp_with_mask(pin.numpy(), K, Rs[0], dst_img=di.numpy(), dst_mask=dm.numpy())
    assert warper.last_host_bands() >= 4
    assert c2 == c0 and np.array_equal(di.numpy(), wi0) and np.array_equal(dm.numpy(), wm0)
    # a source that the caller overwrites right after the call (cv::Mat semantics: consumed when warp() returns)
    src = img.copy()
    _, wi3, _ = warper.warp_with_mask(src, K, Rs[1])
    src[:] = 0
    _, wi4, _ = warper.warp_with_mask(img, K, Rs[1])
    assert np.array_equal(wi3, wi4)
