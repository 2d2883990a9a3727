"""GPU parity of the warper (A1-A8) against the CPU oracle: bit-exact through the C-ABI."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu

CYL, SPH = 0, 1
NEAREST, LINEAR = 0, 1
CONST, REPL, REFLECT, REFLECT101 = 0, 1, 2, 4


def _cams(w, h, f, yaw=0.3):
    return synth.camera_pair(w, h, f, yaw=yaw)


@pytest.mark.parametrize("kind", [CYL, SPH])
@pytest.mark.parametrize("size", [(301, 211), (640, 360)])
def test_camera_roi_maps(gpu, oracle, kind, size):
    w, h = size
    f = 0.8 * w
    K, Rs = _cams(w, h, f)
    warper = (gpu.CylindricalWarper if kind == CYL else gpu.SphericalWarper)().create(f)
    for R in Rs:
        r_kinv, k_rinv = warper.camera(K, R)
        k, rinv, o_rk, o_kr = oracle.camera(K, R)
        assert np.array_equal(r_kinv, o_rk) and np.array_equal(k_rinv, o_kr)
        roi, mm = warper.warpRoi((w, h), K, R, with_minmax=True)
        o_roi, o_mm = oracle.detect_roi(kind, f, K, R, w, h)
        assert tuple(o_roi) == roi
        assert np.array_equal(mm, o_mm), (mm, o_mm)
        roi2, xm, ym = warper.buildMaps((w, h), K, R)
        oxm, oym = oracle.build_maps(kind, f, o_kr, o_roi)
        assert roi2 == roi
        assert np.array_equal(xm, oxm) and np.array_equal(ym, oym)


@pytest.mark.parametrize("kind", [CYL, SPH])
@pytest.mark.parametrize("interp,border", [(LINEAR, REFLECT), (NEAREST, CONST), (LINEAR, CONST), (NEAREST, REFLECT101), (LINEAR, REPL)])
def test_warp_u8_bit_exact(gpu, oracle, kind, interp, border):
    w, h = 333, 217
    f = 260.0
    K, Rs = _cams(w, h, f, yaw=0.45)
    img = synth.make_tile(h, w, 3, noise_only=True)
    warper = (gpu.CylindricalWarper if kind == CYL else gpu.SphericalWarper)().create(f)
    for R in Rs:
        for src in (img, np.ascontiguousarray(img[:, :, 1])):
            corner, dst = warper.warp(src, K, R, interp, border)
            o_corner, o_dst, _ = oracle.warp_u8(kind, f, K, R, src, interp, border)
            assert corner == o_corner
            assert dst.shape == o_dst.shape
            assert np.array_equal(dst, o_dst), np.argwhere(dst != o_dst)[:5]


def test_warp_f32_bit_exact(gpu, oracle):
    w, h = 200, 150
    f = 170.0
    K, Rs = _cams(w, h, f)
    img = synth.make_tile(h, w, 5, noise_only=True).astype(np.float32) * np.float32(1.37)
    warper = gpu.CylindricalWarper().create(f)
    corner, dst = warper.warp(img, K, Rs[0], LINEAR, REFLECT)
    k, rinv, rk, kr = oracle.camera(K, Rs[0])
    roi, _ = oracle.detect_roi(CYL, f, K, Rs[0], w, h)
    xm, ym = oracle.build_maps(CYL, f, kr, roi)
    ref = oracle.remap(img, xm, ym, LINEAR, REFLECT)
    assert corner == (roi[0], roi[1])
    assert np.array_equal(dst, ref)


@pytest.mark.parametrize("out16", [False, True])
def test_warp_with_mask_fused(gpu, oracle, out16):
    """W:229 + W:232 (+ W:294) in one pass == the two separate warps (+ convertTo(CV_16S))."""
    import torch
    w, h = 480, 270
    f = 400.0
    K, Rs = _cams(w, h, f)
    img = synth.make_tile(h, w, 7)
    warper = gpu.CylindricalWarper().create(f)
    t_img = torch.from_numpy(img).cuda()
    corner, wi, wm = warper.warp_with_mask(t_img, K, Rs[1], out16=out16)
    o_corner, o_img, _ = oracle.warp_u8(CYL, f, K, Rs[1], img, LINEAR, REFLECT)
    _, o_mask, _ = oracle.warp_u8(CYL, f, K, Rs[1], np.full((h, w), 255, np.uint8), NEAREST, CONST)
    assert corner == o_corner
    assert np.array_equal(wi.cpu().numpy(), o_img.astype(np.int16) if out16 else o_img)
    assert np.array_equal(wm.cpu().numpy(), o_mask)
    # explicit source mask with holes
    mask = (synth.make_tile(h, w, 9, noise_only=True)[:, :, 0] > 40).astype(np.uint8) * 255
    _, _, wm2 = warper.warp_with_mask(t_img, K, Rs[1], mask=torch.from_numpy(mask).cuda(), out16=out16)
    _, o_mask2, _ = oracle.warp_u8(CYL, f, K, Rs[1], mask, NEAREST, CONST)
    assert np.array_equal(wm2.cpu().numpy(), o_mask2)


def test_warp_errors(gpu):
    warper = gpu.CylindricalWarper().create(100.0)
    K, Rs = _cams(64, 48, 100.0)
    img = np.zeros((48, 64, 3), np.uint8)
    with pytest.raises(gpu.IsxError) as e:
        warper.warp(img, K, Rs[0], LINEAR, REFLECT, dst=np.zeros((5, 5, 3), np.uint8))
    assert e.value.code == 7
    with pytest.raises(gpu.IsxError):
        warper.warp(img.astype(np.int16), K, Rs[0], LINEAR, REFLECT, dst=np.zeros((5, 5, 3), np.int16))
    with pytest.raises(gpu.IsxError):
        gpu.CylindricalWarper().create(-1.0)
