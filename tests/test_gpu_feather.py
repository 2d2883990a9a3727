"""the FeatherBlender every reference demo runs (W:278-281,302,313) and the mask
preparation in front of it (dilate 20x20 + AND, W:286-301), against the CPU oracle — bit-exact."""
import os

import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _mask(rng, h, w, holes=True, p=0.002):
    m = np.full((h, w), 255, np.uint8)
    if holes:
        m[rng.random((h, w)) < p] = 0
        m[: h // 7, : w // 5] = 0
    return m


@pytest.mark.parametrize("ksize", [(20, 20), (3, 3), (5, 2), (1, 9), (40, 7)])
def test_dilate_and(gpu, oracle, ksize):
    import torch
    rng = np.random.default_rng(sum(ksize))
    m = (rng.random((123, 211)) < 0.01).astype(np.uint8) * 255
    other = (rng.random((123, 211)) < 0.7).astype(np.uint8) * 255
    ref = oracle.dilate_rect(m, *ksize)
    assert np.array_equal(gpu.dilate_and(m, *ksize), ref)
    assert np.array_equal(gpu.dilate_and(m, *ksize, other=other), ref & other)
    out = gpu.dilate_and(torch.from_numpy(m).cuda(), *ksize, other=torch.from_numpy(other).cuda())
    assert np.array_equal(out.cpu().numpy(), ref & other)


@pytest.mark.parametrize("sharpness", [0.1, 0.02, 0.5])
@pytest.mark.parametrize("holes", [True, False])
def test_feather_two_tiles_bit_exact(gpu, oracle, sharpness, holes):
    rng = np.random.default_rng(int(sharpness * 100) + holes)
    corners, sizes = [(-4, 6), (70, -2)], [(130, 97), (111, 105)]
    tiles = [(synth.make_tile(h, w, 60 + i, noise_only=True).astype(np.int16), _mask(rng, h, w, holes)) for i, (w, h) in enumerate(sizes)]
    fb = gpu.Blender.createDefault(gpu.Blender.FEATHER, False)
    fb.setSharpness(sharpness)
    ob = oracle.Feather(sharpness)
    fb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        fb.feed(img, mask, c)
        ob.feed(img, mask, c)
    lap, w = fb.level(0)
    d, m = fb.blend()
    od, om = ob.blend()
    assert np.array_equal(m, om)
    assert np.array_equal(d, od), np.argwhere(d != od)[:5]
    assert w.max() <= float(len(tiles)) and w.min() >= 0.0


def test_feather_errors_and_u8_entry(gpu, oracle):
    import torch
    fb = gpu.FeatherBlender(False, 0.1)
    img = synth.make_tile(40, 50, 1)
    mask = np.full((40, 50), 255, np.uint8)
    fb.prepare([(0, 0)], [(50, 40)])
    with pytest.raises(gpu.IsxError) as e:
        fb.feed(img.astype(np.float32), mask, (0, 0))       # CV_Assert(img.type() == CV_16SC3)
    assert e.value.code == 2
    with pytest.raises(gpu.IsxError):
        fb.feed(img.astype(np.int16), mask, (30, 0))        # outside the prepared ROI
    fb.feed_u8(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), (0, 0))
    d, m = fb.blend()
    ob = oracle.Feather(0.1)
    ob.prepare([(0, 0)], [(50, 40)])
    ob.feed(img.astype(np.int16), mask, (0, 0))
    od, om = ob.blend()
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(m.cpu().numpy(), om)
    mb = gpu.MultiBandBlender(False, 3, gpu.PREC_I16)
    with pytest.raises(gpu.IsxError):
        gpu._lib.check(gpu._lib.load().isx_blender_set_sharpness(mb._h, 0.1))   # not a FeatherBlender


def test_reference_blend_stage_on_its_own_artefacts(gpu, oracle):
    """W:278-313 as the demos run it: FeatherBlender(sharpness 0.1), dilate(seam mask, 20x20) & warped mask,
    convertTo(CV_16S), feed, blend — on crops of the reference's committed warped tiles and DP-seam masks."""
    D = np.load(os.path.join(HERE, "golden", "ref_inputs.npz"))
    imgs = [D["img0"], D["img1"]]
    seam = [D["mask0"], D["mask1"]]
    warped = [np.where(im.sum(2) > 0, 255, 0).astype(np.uint8) for im in imgs]     # stand-in for masks_warped: the tile's support
    corners = [tuple(int(v) for v in D["corner0"]), tuple(int(v) for v in D["corner1"])]
    sizes = [(m.shape[1], m.shape[0]) for m in seam]
    fb = gpu.Blender.createDefault(gpu.Blender.FEATHER, False)
    fb.setSharpness(0.1)                                                             # W:280
    ob = oracle.Feather(0.1)
    fb.prepare(corners, sizes)                                                       # W:281
    ob.prepare(corners, sizes)
    for k in range(2):
        img_s = imgs[k].astype(np.int16)                                             # W:294
        mk = gpu.dilate_and(seam[k], 20, 20, other=warped[k])                        # W:295-301
        assert np.array_equal(mk, oracle.dilate_rect(seam[k], 20, 20) & warped[k])
        fb.feed(img_s, mk, corners[k])                                               # W:302
        ob.feed(img_s, mk, corners[k])
    d, m = fb.blend()                                                                # W:313
    od, om = ob.blend()
    assert np.array_equal(m, om) and np.array_equal(d, od)
    assert (m == 255).mean() > 0.5


def test_feather_full_size_4k(gpu, oracle):
    import torch
    h, w = 2160, 3425
    rng = np.random.default_rng(3)
    img = synth.make_tile(h, w, 2).astype(np.int16)
    mask = _mask(rng, h, w, True, p=1e-5)
    fb = gpu.FeatherBlender(False, 0.1)
    ob = oracle.Feather(0.1)
    fb.prepare([(0, 0)], [(w, h)])
    ob.prepare([(0, 0)], [(w, h)])
    fb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), (0, 0))
    ob.feed(img, mask, (0, 0))
    d, m = fb.blend()
    od, om = ob.blend()
    assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)


@pytest.mark.parametrize("shape,off", [((37, 4200), 0), ((70, 9001), 3), ((33, 64), 1), ((130, 4097), 2), ((5, 17), 1)])
def test_weight_map_wide_rows_and_unaligned_views(gpu, oracle, shape, off):
    """createWeightMap on rows wider than one 4096-pixel pass of the row kernel and on device views whose rows start at
    any byte offset: the accumulated weight of a single fed tile IS its weight map."""
    import torch
    h, w = shape
    rng = np.random.default_rng(h * w + off)
    big = np.full((h, w + off), 255, np.uint8)
    big[rng.random(big.shape) < 0.0008] = 0
    big[h // 2, :] = 255                       # a row without any zero
    if h > 8: big[3, w // 3: w // 3 + 900] = 0  # a long run of zeros crossing a chunk boundary on the wide cases
    mask = big[:, off:]
    img = rng.integers(-300, 300, (h, w, 3)).astype(np.int16)
    fb = gpu.FeatherBlender(False, 0.02)
    fb.prepare([(0, 0)], [(w, h)])
    fb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(big).cuda()[:, off:], (0, 0))
    _, wgt = fb.level(0)
    ref = oracle.feather_weight_map(np.ascontiguousarray(mask), 0.02)
    assert np.array_equal(wgt, ref), np.argwhere(wgt != ref)[:5]
    ob = oracle.Feather(0.02)
    ob.prepare([(0, 0)], [(w, h)])
    ob.feed(img, np.ascontiguousarray(mask), (0, 0))
    d, m = fb.blend()
    od, om = ob.blend()
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(m.cpu().numpy(), om)


def test_feather_many_tiles_without_clear(gpu, oracle):
    """More tiles than one Cover holds (8): the accumulators are never memset, uncovered pixels are defined as zero."""
    rng = np.random.default_rng(11)
    n = 11
    sizes = [(40 + 3 * i, 30 + 2 * i) for i in range(n)]
    corners = [(int(rng.integers(-20, 120)), int(rng.integers(-10, 60))) for _ in range(n)]
    fb, ob = gpu.FeatherBlender(False, 0.1), oracle.Feather(0.1)
    for cycle in range(2):                      # second cycle re-uses the (dirty) arena
        fb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        for i, ((w, h), c) in enumerate(zip(sizes, corners)):
            img = synth.make_tile(h, w, 100 + i + cycle, noise_only=True).astype(np.int16)
            mask = _mask(rng, h, w, True, p=0.01)
            fb.feed(img, mask, c)
            ob.feed(img, mask, c)
        d, m = fb.blend()
        od, om = ob.blend()
        assert np.array_equal(m, om) and np.array_equal(d, od)


@pytest.mark.parametrize("shape,ksize,off", [((300, 517), (20, 20), 1), ((129, 65), (33, 33), 0), ((64, 128), (1, 1), 2),
                                             ((257, 300), (2, 31), 3), ((200, 180), (34, 5), 0), ((90, 70), (7, 60), 1)])
def test_dilate_tiles_halo_and_fallback(gpu, oracle, shape, ksize, off):
    """Sizes that straddle the 64 x 128 tile of the fused kernel, unaligned device views, and structuring elements
    beyond 33 (generic two-kernel path)."""
    import torch
    h, w = shape
    rng = np.random.default_rng(h + w + sum(ksize))
    big = rng.integers(0, 256, (h, w + off)).astype(np.uint8)
    big[rng.random(big.shape) < 0.97] = 0
    other = rng.integers(0, 256, (h, w)).astype(np.uint8)
    m = np.ascontiguousarray(big[:, off:])
    ref = oracle.dilate_rect(m, *ksize)
    assert np.array_equal(gpu.dilate_and(m, *ksize), ref)
    out = gpu.dilate_and(torch.from_numpy(big).cuda()[:, off:], *ksize, other=torch.from_numpy(other).cuda())
    assert np.array_equal(out.cpu().numpy(), ref & other)


@pytest.mark.parametrize("n", [2, 5, 11])
def test_feather_deferred_gather_equals_eager_and_oracle(gpu, oracle, n):
    """isx_blender_set_deferred_level0 on a FeatherBlender: feed() builds only the weight maps, blend() gathers over the
    recorded tiles (no accumulators); beyond 8 tiles the cycle falls back to the eager path. Device tiles, u8 entry."""
    import torch
    rng = np.random.default_rng(n)
    sizes = [(90 + 7 * i, 60 + 5 * i) for i in range(n)]
    corners = [(int(rng.integers(-30, 150)), int(rng.integers(-20, 80))) for _ in range(n)]
    imgs = [synth.make_tile(h, w, 40 + i, noise_only=True) for i, (w, h) in enumerate(sizes)]
    masks = [_mask(rng, h, w, True, p=0.004) for (w, h) in sizes]
    timgs = [torch.from_numpy(a).cuda() for a in imgs]
    tmasks = [torch.from_numpy(a).cuda() for a in masks]
    ob = oracle.Feather(0.05)
    ob.prepare(corners, sizes)
    for im, mk, c in zip(imgs, masks, corners):
        ob.feed(im.astype(np.int16), mk, c)
    od, om = ob.blend()
    outs = []
    for deferred in (False, True):
        fb = gpu.FeatherBlender(False, 0.05)
        fb.set_deferred_level0(deferred)
        for cycle in range(2):
            fb.prepare(corners, sizes)
            for im, mk, c in zip(timgs, tmasks, corners):
                fb.feed_u8(im, mk, c)
            d, m = fb.blend()
        outs.append((d.cpu().numpy(), m.cpu().numpy()))
        assert np.array_equal(outs[-1][1], om) and np.array_equal(outs[-1][0], od)
    # host mats in a deferred cycle: every recorded tile is staged into a buffer of its own (CV_16SC3 entry)
    fb = gpu.FeatherBlender(False, 0.05)
    fb.set_deferred_level0(True)
    fb.prepare(corners, sizes)
    for im, mk, c in zip(imgs, masks, corners):
        fb.feed(im.astype(np.int16), mk, c)
    d, m = fb.blend()
    assert np.array_equal(m, om) and np.array_equal(d, od)


def test_feather_deferred_uses_gather_and_flushes(gpu, oracle):
    import torch
    lib = gpu._lib.load()
    rng = np.random.default_rng(5)
    sizes, corners = [(120, 80), (100, 90)], [(0, 0), (70, 10)]
    imgs = [synth.make_tile(h, w, 3 + i, noise_only=True) for i, (w, h) in enumerate(sizes)]
    masks = [_mask(rng, h, w, True, p=0.004) for (w, h) in sizes]
    ob = oracle.Feather(0.1)
    ob.prepare(corners, sizes)
    for im, mk, c in zip(imgs, masks, corners):
        ob.feed(im.astype(np.int16), mk, c)
    od, om = ob.blend()
    fb = gpu.FeatherBlender(False, 0.1)
    fb.set_deferred_level0(True)
    # (a) pure deferred cycle: one gather launch, no accumulate / normalise launches
    lib.isx_profile_enable(1); lib.isx_profile_reset()
    fb.prepare(corners, sizes)
    keep = [(torch.from_numpy(im).cuda(), torch.from_numpy(mk).cuda()) for im, mk in zip(imgs, masks)]   # the contract: alive until blend()
    for (im, mk), c in zip(keep, corners):
        fb.feed_u8(im, mk, c)
    d, m = fb.blend()
    ent = gpu._lib.profile_entries()
    lib.isx_profile_enable(0)
    assert ent["feather_gather"]["launches"] == 1 and "feather_acc" not in ent and "feather_blend" not in ent
    assert np.array_equal(d.cpu().numpy(), od) and np.array_equal(m.cpu().numpy(), om)
    # (b) a second source type in the same cycle (CV_16SC3 after CV_8UC3) and level introspection flush to the eager path
    fb.prepare(corners, sizes)
    fb.feed_u8(imgs[0], masks[0], corners[0])
    fb.feed(imgs[1].astype(np.int16), masks[1], corners[1])
    _, w = fb.level(0)
    assert w.max() > 0
    d, m = fb.blend()
    assert np.array_equal(d, od) and np.array_equal(m, om)


@pytest.mark.parametrize("gain", [1.0, 0.5, 0.9731, 1.0379, 2.5, -1.0, 1e8, float("inf"), float("nan")])
def test_gain_apply(gpu, oracle, gain):
    """compensator->apply (W:241-244) in place, host mats, pitched / unaligned device views, 1 and 3 channels."""
    import torch
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (67, 131, 3)).astype(np.uint8)
    ref = oracle.gain_apply(img, gain)
    a = img.copy()
    assert gpu.gain_apply(a, gain) is a and np.array_equal(a, ref)
    t = torch.from_numpy(img).cuda()
    gpu.gain_apply(t, gain)
    assert np.array_equal(t.cpu().numpy(), ref)
    big = torch.from_numpy(rng.integers(0, 256, (67, 140, 3)).astype(np.uint8)).cuda()     # a view: rows start at odd addresses
    view = big[:, 3:134]
    before = big.cpu().numpy().copy()
    gpu.gain_apply(view, gain)
    after = big.cpu().numpy()
    assert np.array_equal(after[:, 3:134], oracle.gain_apply(before[:, 3:134], gain))
    assert np.array_equal(after[:, :3], before[:, :3]) and np.array_equal(after[:, 134:], before[:, 134:])   # nothing outside the view
    pitched = torch.zeros((50, 256), dtype=torch.uint8, device="cuda")                      # aligned pitch, row length 201 (dword tail)
    m = rng.integers(0, 256, (50, 201)).astype(np.uint8)
    pitched[:, :201] = torch.from_numpy(m).cuda()
    gpu.gain_apply(pitched[:, :201], gain)
    out = pitched.cpu().numpy()
    assert np.array_equal(out[:, :201], oracle.gain_apply(m, gain)) and not out[:, 201:].any()


def test_feather_stage_agrees_with_the_references_pano(gpu, oracle):
    """The demo's final stage (S:1236-1283) through the HIP library on the reference's own artefacts: identical to the oracle's
    result, and as close to the committed pano.jpg in the seam band as JPEG noise allows (see tests/test_ref_artifact.py)."""
    import sys
    sys.path.insert(0, HERE)
    from test_ref_artifact import _demo_blend, _psnr_in_zone, dpseam_case
    a = np.load(os.path.join(HERE, "golden", "ref_dpseam_artifact.npz"))
    c = dpseam_case()
    got = _demo_blend(lambda s: gpu.FeatherBlender(False, s), lambda m, kw, kh: gpu.dilate_and(m, kw, kh), None, c, 0.1, True)
    ref = _demo_blend(lambda s: oracle.Feather(s), oracle.dilate_rect, None, c, 0.1, True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert _psnr_in_zone(got[0], got[1], a) > 42.0
