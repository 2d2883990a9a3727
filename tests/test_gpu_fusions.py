"""SURVEY N3's fusions (round 4): the gain of W:241-244 folded into the fused tile warp (isx_warper_set_gain) and the mask preparation of
W:286-301 folded into the feed (isx_blender_feed_dilated) - each bit for bit the separate stages (oracle)."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


@pytest.mark.parametrize("kind", ["cylindrical", "spherical"])
def test_gain_folded_into_the_tile_warp(gpu, oracle, kind):
    import torch
    O = oracle
    pk = O.CYL if kind == "cylindrical" else O.SPH
    W, H, F = 700, 420, 520.0
    K, Rs = synth.camera_pair(W, H, F, yaw=0.5, pitch=0.05, roll=0.08)
    img = synth.make_tile(H, W, 3, noise_only=True)
    warper = (gpu.CylindricalWarper if kind == "cylindrical" else gpu.SphericalWarper)().create(F)
    oc, owi, roi = O.warp_u8(pk, F, K, Rs[0], img, O.LINEAR, O.BORDER_REFLECT)
    _, owm, _ = O.warp_u8(pk, F, K, Rs[0], np.full((H, W), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
    for gain in (0.98872, 0.5, 1.3, 2.7, 0.0, 1.0, float("nan")):
        want = O.gain_apply(owi, gain)           # compensator->apply on the warped image (W:241-244)
        warper.set_gain(gain)
        for dev in (False, True):
            src = torch.from_numpy(img).cuda() if dev else img
            c, wi, wm = warper.warp_with_mask(src, K, Rs[0])
            assert c == oc
            assert np.array_equal(_np(wm), owm), gain
            assert np.array_equal(_np(wi), want), (gain, dev)
            c, wi16, wm = warper.warp_with_mask(src, K, Rs[0], out16=True)
            assert np.array_equal(_np(wi16), want.astype(np.int16)), (gain, dev, "out16")
        # the planned (sync-free) entry
        h, w = owm.shape
        di = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
        dm = torch.empty((h, w), dtype=torch.uint8, device="cuda")
        warper.warp_with_mask_planned(torch.from_numpy(img).cuda(), K, Rs[0], tuple(int(v) for v in roi), di, dm)
        assert warper.plan_status() == 0
        assert np.array_equal(di.cpu().numpy(), want) and np.array_equal(dm.cpu().numpy(), owm)
    warper.set_gain(1.0)
    c, wi, wm = warper.warp_with_mask(img, K, Rs[0])
    assert np.array_equal(wi, owi)
    # a caller-supplied source mask takes the other kernel: the gain is refused there, not silently dropped
    warper.set_gain(1.1)
    with pytest.raises(gpu.IsxError) as e:
        warper.warp_with_mask(img, K, Rs[0], mask=np.full((H, W), 255, np.uint8))
    assert e.value.code == 6


@pytest.mark.parametrize("element", [(20, 20), (1, 1), (33, 7), (5, 33)])
def test_feed_dilated_is_dilate_and_then_feed(gpu, oracle, element):
    """W:286-302: every blender type and cycle, host and device mats; the seam masks are thin random stripes, the warped masks have holes."""
    import torch
    kw, kh = element
    rng = np.random.default_rng(kw * 100 + kh)
    corners, sizes = [(0, 0), (180, 11)], [(300, 200), (290, 190)]
    imgs, seams, wmasks, fed = [], [], [], []
    for (w, h) in sizes:
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.int64).astype(np.int16))
        s = np.zeros((h, w), np.uint8)
        s[rng.random((h, w)) < 0.002] = 255
        s[:, w // 3:w // 2] = 255
        wm = np.full((h, w), 255, np.uint8)
        wm[rng.random((h, w)) < 0.1] = 0
        seams.append(s); wmasks.append(wm)
        fed.append(oracle.dilate_rect(s, kw, kh) & wm)          # dilate(masks_seam, element) & masks_warped  W:295-299
    def run_oracle(ob):
        ob.prepare(corners, sizes)
        for i in range(2):
            ob.feed(imgs[i], fed[i], corners[i])
    ob = oracle.MultiBand(4, 0); run_oracle(ob); want_mb = ob.blend(False)
    of = oracle.Feather(0.1); run_oracle(of); want_fb = of.blend()
    on = oracle.NoBlend(); run_oracle(on); want_nb = on.blend()
    cases = [("multiband eager", lambda: gpu.MultiBandBlender(False, 4, 0), False, want_mb),
             ("multiband deferred", lambda: gpu.MultiBandBlender(False, 4, 0), True, want_mb),
             ("multiband deferred copy", lambda: gpu.MultiBandBlender(False, 4, 0), "copy", want_mb),
             ("feather", lambda: gpu.FeatherBlender(False, 0.1), False, want_fb),
             ("feather deferred", lambda: gpu.FeatherBlender(False, 0.1), True, want_fb),
             ("no", lambda: gpu.NoBlender(), None, want_nb)]
    for name, make, mode, (od, om) in cases:
        for dev in (False, True):
            b = make()
            if mode is not None:
                b.set_deferred_level0(mode)
            b.prepare(corners, sizes)
            keep = []
            for i in range(2):
                if dev:
                    t = [torch.from_numpy(a).cuda() for a in (imgs[i], seams[i], wmasks[i])]
                    keep.append(t)
                    b.feed_dilated(t[0], t[1], t[2], kw, kh, corners[i])
                    if mode == "copy":
                        t[1].fill_(7), t[2].fill_(9)            # the masks are consumed by the call in every mode
                else:
                    b.feed_dilated(imgs[i], seams[i], wmasks[i], kw, kh, corners[i])
            d, m = b.blend()
            assert np.array_equal(_np(m), om), (name, dev)
            assert np.array_equal(_np(d), od), (name, dev)
    b = gpu.MultiBandBlender(False, 4, 0)
    b.prepare(corners, sizes)
    with pytest.raises(gpu.IsxError) as e:
        b.feed_dilated(imgs[0], seams[0], wmasks[0], 34, 3, corners[0])
    assert e.value.code == 6
    # CV_8UC3 tiles take the fused conversion, as feed() does
    b = gpu.MultiBandBlender(False, 4, 0)
    b.set_deferred_level0(True)
    b.prepare(corners, sizes)
    u8 = [np.clip(im, 0, 255).astype(np.uint8) for im in imgs]
    ob = oracle.MultiBand(4, 0)
    ob.prepare(corners, sizes)
    for i in range(2):
        b.feed_dilated(u8[i], seams[i], wmasks[i], kw, kh, corners[i])
        ob.feed(u8[i].astype(np.int16), fed[i], corners[i])
    d, m = b.blend()
    od, om = ob.blend(False)
    assert np.array_equal(m, om) and np.array_equal(d, od)
