"""The reference demo's whole post-registration stage on one pair, every step through the HIP library and compared with the
same chain of oracle steps: warp image + mask (W:223-233), gain apply (W:241-244), convertTo(CV_32F) + DP seam finder
(W:253-262 / S:1188-1192), dilate 20x20 & warped mask (W:286-301), convertTo(CV_16S), FeatherBlender 0.1 (W:278-313) and the
multi-band alternative (W:271-273)."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("blender", ["feather", "multiband"])
def test_demo_stage_end_to_end(gpu, oracle, blender):
    from oracle.dpseam_np import DpSeamFinder as OracleFinder
    W, H, F = 520, 360, 420.0
    K, Rs = synth.camera_pair(W, H, F, yaw=0.22)
    imgs = [synth.make_tile(H, W, 70 + i) for i in range(2)]
    gains = [1.0379, 0.9731]
    warper = gpu.CylindricalWarper().create(F)
    g_corners, g_img, g_mask, o_corners, o_img, o_mask = [], [], [], [], [], []
    for i in range(2):
        c, wi, wm = warper.warp_with_mask(imgs[i], K, Rs[i])                            # W:229 + W:232
        gpu.gain_apply(wi, gains[i])                                                    # W:241-244
        g_corners.append(tuple(c)); g_img.append(wi); g_mask.append(wm)
        oc, oi, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], imgs[i], 1, 2)
        _, om, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], np.full((H, W), 255, np.uint8), 0, 0)
        o_corners.append(oc); o_img.append(oracle.gain_apply(oi, gains[i])); o_mask.append(om)
        assert g_corners[i] == oc and np.array_equal(g_img[i], o_img[i]) and np.array_equal(g_mask[i], om)
    g_seam = [m.copy() for m in g_mask]                                                 # masks_seam = copies of masks_warped, W:247-249
    gpu.DpSeamFinder().find([a.astype(np.float32) for a in g_img], g_corners, g_seam)   # W:259-262
    o_seam = [m.copy() for m in o_mask]
    OracleFinder().find([a.astype(np.float32) for a in o_img], o_corners, o_seam)
    assert all(np.array_equal(a, b) for a, b in zip(g_seam, o_seam)) and (g_seam[0] != g_mask[0]).any()
    sizes = [(a.shape[1], a.shape[0]) for a in g_img]
    if blender == "feather":
        gb, ob = gpu.FeatherBlender(False, 0.1), oracle.Feather(0.1)                    # W:278-280
    else:
        gb, ob = gpu.MultiBandBlender(False, 4, gpu.PREC_I16), oracle.MultiBand(4, oracle.I16)   # W:271-273
    gb.prepare(g_corners, sizes); ob.prepare(o_corners, sizes)                          # W:281
    for i in range(2):
        mk = gpu.dilate_and(g_seam[i], 20, 20, other=g_mask[i])                         # W:295-301
        omk = oracle.dilate_rect(o_seam[i], 20, 20) & o_mask[i]
        assert np.array_equal(mk, omk)
        gb.feed(g_img[i].astype(np.int16), mk, g_corners[i])                            # W:294, W:302
        ob.feed(o_img[i].astype(np.int16), omk, o_corners[i])
    res, rmask = gb.blend()                                                             # W:313
    ores, omask = ob.blend() if blender == "feather" else ob.blend(False)
    assert np.array_equal(rmask, omask) and np.array_equal(res, ores)
    assert (rmask == 255).mean() > 0.8
