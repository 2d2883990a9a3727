"""GPU parity of the multi-band blender (A9-A12) and the linear pair blend (A13) against the oracle."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu

I16, F32, F16 = 0, 1, 2


def _tiles(rng, sizes, noise=True):
    out = []
    for i, (w, h) in enumerate(sizes):
        img = synth.make_tile(h, w, 11 + i, noise_only=noise).astype(np.int16)
        mask = (rng.random((h, w)) > 0.25).astype(np.uint8) * 255
        out.append((img, mask))
    return out


def _compare_levels(mb, ob, prec):
    for i in range(ob.num_bands + 1):
        lap, w = mb.level(i)
        olap, ow = ob.level(i)
        assert lap.shape == olap.shape
        assert np.array_equal(w, ow), ("weight level", i, np.abs(w - ow).max())
        assert np.array_equal(lap, olap), ("laplacian level", i, prec, np.argwhere(lap != olap)[:4])


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("bands", [0, 1, 3, 5])
def test_multiband_two_tiles_bit_exact(gpu, oracle, prec, bands):
    rng = np.random.default_rng(100 + bands)
    corners = [(-7, 4), (61, -3)]
    sizes = [(101, 83), (97, 90)]
    tiles = _tiles(rng, sizes)
    mb = gpu.MultiBandBlender(False, bands, prec)
    ob = oracle.MultiBand(bands, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    assert mb.numBands() == ob.num_bands
    for (img, mask), c in zip(tiles, corners):
        mb.feed(img, mask, c)
        ob.feed(img, mask, c)
    _compare_levels(mb, ob, prec)
    of32 = prec != I16
    dst, dmask = mb.blend(out_f32=of32)
    odst, omask = ob.blend(of32)
    assert np.array_equal(dmask, omask)
    assert np.array_equal(dst, odst), np.argwhere(dst != odst)[:4]


@pytest.mark.parametrize("prec", [I16, F32])
def test_multiband_three_tiles_ragged(gpu, oracle, prec):
    """three overlapping tiles, sizes not multiples of anything, one fully inside another's gap zone"""
    rng = np.random.default_rng(5)
    corners = [(0, 0), (150, 20), (70, -40)]
    sizes = [(200, 131), (171, 150), (66, 59)]
    tiles = _tiles(rng, sizes)
    mb = gpu.MultiBandBlender(False, 4, prec)
    ob = oracle.MultiBand(4, prec)
    mb.prepare(corners, sizes)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        mb.feed(img, mask, c)
        ob.feed(img, mask, c)
    _compare_levels(mb, ob, prec)
    dst, dmask = mb.blend(out_f32=False)
    odst, omask = ob.blend(False)
    assert np.array_equal(dmask, omask) and np.array_equal(dst, odst)


def test_multiband_tiny_and_band_clamp(gpu, oracle):
    """num_bands is clamped to ceil(log2(max(w, h))); the top level becomes 1 pixel wide (pyrUp n == 1)."""
    rng = np.random.default_rng(9)
    corners, sizes = [(0, 0), (9, 2)], [(20, 9), (14, 11)]
    tiles = _tiles(rng, sizes)
    for prec in (I16, F32):
        mb = gpu.MultiBandBlender(False, 7, prec)
        ob = oracle.MultiBand(7, prec)
        mb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        assert mb.numBands() == ob.num_bands == 5
        for (img, mask), c in zip(tiles, corners):
            mb.feed(img, mask, c)
            ob.feed(img, mask, c)
        _compare_levels(mb, ob, prec)
        d, m = mb.blend()
        od, om = ob.blend(False)
        assert np.array_equal(d, od) and np.array_equal(m, om)


def test_feed_u8_equals_convert_then_feed(gpu, oracle):
    import torch
    rng = np.random.default_rng(3)
    corners, sizes = [(0, 0), (50, 5)], [(90, 70), (88, 64)]
    tiles = _tiles(rng, sizes)
    res = []
    for use_u8 in (False, True):
        mb = gpu.MultiBandBlender(False, 3, F32)
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            if use_u8:
                mb.feed_u8(torch.from_numpy(img.astype(np.uint8)).cuda(), torch.from_numpy(mask).cuda(), c)
            else:
                mb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), c)
        d, m = mb.blend(out_f32=True)
        res.append((d.cpu().numpy(), m.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    ob = oracle.MultiBand(3, F32)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    od, om = ob.blend(True)
    assert np.array_equal(res[0][0], od) and np.array_equal(res[0][1], om)


def test_blender_errors(gpu):
    mb = gpu.MultiBandBlender(False, 3, I16)
    img = np.zeros((10, 10, 3), np.int16)
    mask = np.full((10, 10), 255, np.uint8)
    with pytest.raises(gpu.IsxError) as e:
        mb.feed(img, mask, (0, 0))
    assert e.value.code == 3  # feed before prepare
    mb.prepare([(0, 0)], [(10, 10)])
    with pytest.raises(gpu.IsxError) as e:
        mb.feed(img.astype(np.float32), mask, (0, 0))
    assert e.value.code == 2  # CV_32FC3 in I16 precision
    mb.feed(img, mask, (0, 0))
    mb.blend()
    with pytest.raises(gpu.IsxError) as e:
        mb.blend()
    assert e.value.code == 3  # blend() released the pyramids
    with pytest.raises(gpu.IsxError):
        gpu.Blender.createDefault(3)     # NO (0), FEATHER (1), MULTI_BAND (2) exist (Blender::NO: tests/test_gpu_s16_tiles.py)


@pytest.mark.parametrize("dy", [3, -5, 0])
def test_linear_pair_blend(gpu, oracle, dy):
    import ctypes as C
    from imagestitch_amd import _lib
    rng = np.random.default_rng(17)
    h1, w1, h2, w2 = 120, 160, 123, 150
    img1 = rng.random((h1, w1, 3)).astype(np.float32) * 255
    img2 = rng.random((h2, w2, 3)).astype(np.float32) * 255
    # black corners so that all four overlap classes occur
    img1[:15, -25:] = 3.0
    img2[-20:, :18] = 2.0
    img2[40:50, 20:30] = 1.0
    tl1, tl2 = (10, 20), (10 + 95, 20 + dy)
    rc, opano, oseam = oracle.blend_pair_linear(img1, img2, tl1, tl2)
    assert rc == 0
    lib = _lib.load()
    pano = np.empty_like(opano)
    seam = np.zeros(opano.shape[0], np.int32)
    m1, m2, mp = _lib.as_mat(img1), _lib.as_mat(img2), _lib.as_mat(pano)
    _lib.check(lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), tl1[0], tl1[1], tl2[0], tl2[1], C.byref(mp),
                                         seam.ctypes.data_as(_lib._IP), 0, None))
    assert np.array_equal(seam, oseam)
    assert np.array_equal(pano, opano, equal_nan=True), np.argwhere(pano != opano)[:5]


def test_linear_pair_blend_4k(gpu, oracle):
    """A13 at the size of a warped 4K pair (3416 x 2170 tiles, 1257-column overlap = 20 of k_lin_seam's 64-column LDS windows, the
    seam free to wander across them), device mats."""
    import ctypes as C
    import torch
    from imagestitch_amd import _lib
    rng = np.random.default_rng(41)
    h1, w1, h2, w2 = 2170, 3416, 2166, 3416
    yy, xx = np.mgrid[0:h1, 0:w1].astype(np.float32)
    base = 120.0 + 60.0 * np.sin(xx / 97.0) * np.cos(yy / 61.0)
    img1 = np.clip(base[..., None] + rng.normal(0, 1, (h1, w1, 3)), 0, 255).astype(np.float32)
    shifted = np.roll(base, -2159, axis=1)[:h2, :w2]
    # the two tiles agree along a slanted line through the overlap: the greedy seam (B:268-307) follows it from column 628 to 1167
    valley = np.minimum(0.8 * np.abs(xx[:h2, :w2] - (628.0 + 0.25 * yy[:h2, :w2])), 60.0)
    valley[:, 1257:] = 0
    img2 = np.clip((shifted + valley)[..., None] + rng.normal(0, 1, (h2, w2, 3)), 0, 255).astype(np.float32)
    img1[:300, -500:] = 2.0; img2[-400:, :350] = 1.0; img2[900:1000, 200:300] = 3.0
    tl1, tl2 = (-2788, -1085), (-2788 + 2159, -1085 + 4)
    rc, opano, oseam = oracle.blend_pair_linear(img1, img2, tl1, tl2)
    assert rc == 0 and (oseam.max() - oseam.min()) > 500      # the greedy seam crosses eight of the 64-column windows
    lib = _lib.load()
    t1, t2 = torch.from_numpy(img1).cuda(), torch.from_numpy(img2).cuda()
    pano = torch.empty(opano.shape, dtype=torch.float32, device="cuda")
    seam = np.zeros(opano.shape[0], np.int32)
    m1, m2, mp = _lib.as_mat(t1), _lib.as_mat(t2), _lib.as_mat(pano)
    _lib.check(lib.isx_blend_pair_linear(C.byref(m1), C.byref(m2), tl1[0], tl1[1], tl2[0], tl2[1], C.byref(mp), seam.ctypes.data_as(_lib._IP), 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(seam, oseam)
    assert np.array_equal(pano.cpu().numpy(), opano, equal_nan=True)


def test_blend_on_reference_artefact_crops(gpu, oracle):
    """Inputs: crops of the reference's committed images_warped_f[*].bmp and its real DP-seam masks
    mask_seam[*].bmp (S:1195-1198); blender configured as the reference does (setNumBands(4), W:273)."""
    import os
    D = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_inputs.npz"))
    imgs = [D["img0"].astype(np.int16), D["img1"].astype(np.int16)]       # convertTo(CV_16S), W:294
    masks = [D["mask0"], D["mask1"]]
    corners = [tuple(int(v) for v in D["corner0"]), tuple(int(v) for v in D["corner1"])]
    sizes = [(m.shape[1], m.shape[0]) for m in masks]
    for prec in (I16, F32):
        mb = gpu.MultiBandBlender(False, 5, prec)
        mb.setNumBands(4)
        ob = oracle.MultiBand(4, prec)
        mb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        for i in range(2):
            mb.feed(imgs[i], masks[i], corners[i])
            ob.feed(imgs[i], masks[i], corners[i])
        d, m = mb.blend()
        od, om = ob.blend(False)
        assert np.array_equal(m, om) and np.array_equal(d, od)
        assert m.max() == 255 and 0 < (m == 255).mean() <= 1.0


def test_full_size_4k_pair(gpu, oracle):
    """BASELINE config 2 at full size: 2 x 3840x2160, cylindrical f=3000, 5 bands.
    Warp: bit-exact vs the oracle.  Blend: bit-exact vs the oracle in I16 (OpenCV's arithmetic) and F32,
    plus size-independent properties (mask = union of the fed masks, zero outside, tile returned when
    both inputs are the same image)."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 3840, 2160, 3000.0
    K, Rs = synth.camera_pair(W, H, F)
    imgs = [synth.make_tile(H, W, i) for i in range(2)]
    dev = torch.device("cuda:0")
    for prec in (gpu.PREC_I16, gpu.PREC_F32):
        # the fp32 blender writes (and is compared as) CV_32FC3: after saturate_cast<short> a sub-LSB difference could not show
        f32 = prec == gpu.PREC_F32
        ps = PairStitcher([torch.from_numpy(i).to(dev) for i in imgs], K, Rs, F, "cylindrical", 5, prec, 0, None, "float32" if f32 else "int16")
        out, omask = ps.step()
        out, omask = out.cpu().numpy(), omask.cpu().numpy()
        if prec == gpu.PREC_I16:
            o_warp = []
            for i in range(2):
                c, wi, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], imgs[i], 1, 2)
                _, wm, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], np.full((H, W), 255, np.uint8), 0, 0)
                assert c == ps.corners[i]
                assert np.array_equal(ps.warped[i].cpu().numpy(), wi), "4K warp differs"
                assert np.array_equal(ps.wmasks[i].cpu().numpy(), wm), "4K warped mask differs"
                o_warp.append(wi)
        seam = [s.cpu().numpy() for s in ps.seam]
        ob = oracle.MultiBand(5, prec)
        ob.prepare(ps.corners, ps.sizes)
        for i in range(2):
            ob.feed(o_warp[i].astype(np.int16), seam[i], ps.corners[i])
        od, om = ob.blend(f32)
        assert np.array_equal(omask, om)
        assert out.dtype == od.dtype == (np.float32 if f32 else np.int16)
        assert np.array_equal(out, od), "4K blend differs (precision %d)" % prec
        # properties
        union = np.zeros_like(omask)
        x0 = min(c[0] for c in ps.corners); y0 = min(c[1] for c in ps.corners)
        for i in range(2):
            cx, cy = ps.corners[i][0] - x0, ps.corners[i][1] - y0
            h, w = seam[i].shape
            union[cy:cy + h, cx:cx + w] |= seam[i]
        assert np.array_equal(omask, union)
        assert np.all(out[omask == 0] == 0)
        del ps
    torch.cuda.empty_cache()


def test_more_tiles_than_cover_slots(gpu, oracle):
    """11 tiles: after MAX_COVER (8) feeds the blender clears the uncovered area once and falls back to
    plain read-modify-write; results must not change."""
    rng = np.random.default_rng(21)
    corners = [(37 * i, (i % 3) * 11 - 5) for i in range(11)]
    sizes = [(64 + (i % 4) * 5, 48 + (i % 5) * 3) for i in range(11)]
    tiles = _tiles(rng, sizes)
    for prec in (I16, F32):
        mb = gpu.MultiBandBlender(False, 3, prec)
        ob = oracle.MultiBand(3, prec)
        mb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            mb.feed(img, mask, c)
            ob.feed(img, mask, c)
        _compare_levels(mb, ob, prec)
        d, m = mb.blend()
        od, om = ob.blend(False)
        assert np.array_equal(m, om) and np.array_equal(d, od)


def test_planned_pipeline_matches_sync_pipeline(gpu):
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 960, 540, 700.0
    K, Rs = synth.camera_pair(W, H, F)
    dev = torch.device("cuda:0")
    imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
    ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16")
    a, am = [t.clone() for t in ps.step_sync()]
    b, bm = [t.clone() for t in ps.step()]
    assert ps.check_plan() == 0
    assert torch.equal(a, b) and torch.equal(am, bm)
    # a wrong plan is detected on the device
    ps.rois[0] = (ps.rois[0][0] + 1,) + tuple(ps.rois[0][1:])
    ps.warper.warp_with_mask_planned(imgs[0], K, Rs[0], ps.rois[0], ps.warped[0][:, :-1], ps.wmasks[0][:, :-1])
    with pytest.raises(gpu.IsxError) as e:
        ps.check_plan()
    assert e.value.code == 8


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("ntiles", [1, 2, 3, 11])
def test_deferred_level0_identical_to_eager(gpu, oracle, prec, ntiles):
    """isx_blender_set_deferred_level0: same results as the eager path and as the oracle; host mats,
    device mats, level introspection in the middle (forces the eager flush), > 8 tiles (falls back)."""
    import torch
    rng = np.random.default_rng(77 + ntiles)
    corners = [(41 * i - 9, (i % 3) * 13 - 6) for i in range(ntiles)]
    sizes = [(70 + (i % 4) * 9, 55 + (i % 5) * 4) for i in range(ntiles)]
    tiles = _tiles(rng, sizes)
    ob = oracle.MultiBand(4, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    od, om = ob.blend(False)
    for variant in ("host", "device", "peek"):
        mb = gpu.MultiBandBlender(False, 4, prec)
        mb.set_deferred_level0(True)
        mb.prepare(corners, sizes)
        keep = []
        for k, ((img, mask), c) in enumerate(zip(tiles, corners)):
            if variant == "host":
                mb.feed(img.copy(), mask.copy(), c)         # temporaries: freed right after the call
            else:
                ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
                keep.append((ti, tm))
                mb.feed(ti, tm, c)
            if variant == "peek" and k == 0:
                lap, w = mb.level(0)                          # flushes the deferred tile eagerly
                assert lap.shape[2] == 3
        d, m = mb.blend()
        d = d.cpu().numpy() if hasattr(d, "cpu") else d
        m = m.cpu().numpy() if hasattr(m, "cpu") else m
        assert np.array_equal(m, om), variant
        assert np.array_equal(d, od), (variant, np.argwhere(d != od)[:4])
    # second cycle on the same blender object re-uses the per-tile arenas
    mb.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        mb.feed(img, mask, c)
    d, m = mb.blend()
    assert np.array_equal(d, od) and np.array_equal(m, om)


def test_pipeline_variants_agree(gpu):
    """planned step == interleaved (side-stream chains) step == hipGraph replay == eager (non-deferred) step, wherever the
    ROI verification scans are scheduled (after the last warp / behind a marked launch inside blend())"""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 1024, 600, 800.0
    K, Rs = synth.camera_pair(W, H, F)
    dev = torch.device("cuda:0")
    imgs = [torch.from_numpy(synth.make_tile(H, W, i)).to(dev) for i in range(2)]
    ref = None
    for kw in (dict(deferred=False), dict(deferred=True), dict(deferred=True, interleave=True), dict(verify_at=-1), dict(verify_at=0),
               dict(verify_at=9)):
        ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16", **kw)
        for _ in range(2):
            a, am = [t.clone() for t in ps.step()]
        assert ps.check_plan() == 0
        if ref is None:
            ref = (a, am)
        assert torch.equal(a, ref[0]) and torch.equal(am, ref[1]), kw
    ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16")
    ps.capture()
    for _ in range(3):
        g, gm = ps.replay()
    torch.cuda.synchronize()
    assert torch.equal(g, ref[0]) and torch.equal(gm, ref[1])
    assert ps.check_plan() == 0
    # round 4: a border-scan verification runs BESIDE the graph (queued from the rig per replay, isx_warper_queue_verify): it still sees a
    # plan that no longer fits the rig - here the rig's second camera turns a little after the capture
    assert ps._verify_outside
    from imagestitch_amd import synth as _s
    _, Rs2 = _s.camera_pair(W, H, F, yaw=0.37)
    ps.Rs = [Rs[0], Rs2[1]]
    ps.replay()
    with pytest.raises(gpu.IsxError) as e:
        ps.check_plan()
    assert e.value.code == 8       # ISX_ERR_PLAN
    # ... and the same stitcher with the verification forked inside the graph (the former placement) gives the same mosaic
    import os
    os.environ["ISX_GRAPH_VERIFY_INSIDE"] = "1"
    try:
        pi = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16")
        pi.capture()
        assert not pi._verify_outside
        for _ in range(2):
            g2, gm2 = pi.replay()
        torch.cuda.synchronize()
        assert torch.equal(g2, ref[0]) and torch.equal(gm2, ref[1]) and pi.check_plan() == 0
    finally:
        del os.environ["ISX_GRAPH_VERIFY_INSIDE"]


@pytest.mark.parametrize("prec", [I16, F32])
def test_blend_u8_output_is_convertTo(gpu, oracle, prec):
    """dst CV_8UC3 == blend to CV_16SC3 then saturate_cast<uchar> (result.convertTo(CV_8U))"""
    rng = np.random.default_rng(31)
    corners, sizes = [(0, 0), (60, 4)], [(100, 80), (90, 77)]
    tiles = _tiles(rng, sizes)
    outs = []
    for u8 in (False, True):
        mb = gpu.MultiBandBlender(False, 4, prec)
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            mb.feed((img * 3 - 200).astype(np.int16), mask, c)      # values outside [0,255] exercise the saturation
        outs.append(mb.blend(out_u8=u8))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(np.clip(outs[0][0], 0, 255).astype(np.uint8), outs[1][0])


@pytest.mark.parametrize("prec", [0, 1])
def test_deferred_copy_mode_keeps_opencvs_feed_contract(gpu, oracle, prec):
    """isx_blender_set_deferred_level0(b, 2): feed() takes private copies of the device mats, so the caller may overwrite
    them right after feed() (the reference releases the fed mats before blend(), W:305-308); results as the oracle's."""
    import torch
    rng = np.random.default_rng(91)
    ntiles = 3
    corners = [(37 * i - 5, (i % 2) * 11 - 4) for i in range(ntiles)]
    sizes = [(66 + i * 7, 50 + i * 5) for i in range(ntiles)]
    tiles = _tiles(rng, sizes)
    ob = oracle.MultiBand(4, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    od, om = ob.blend(False)
    mb = gpu.MultiBandBlender(False, 4, prec)
    mb.set_deferred_level0("copy")
    for cycle in range(2):
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            mb.feed(ti, tm, c)
            ti.fill_(-12345 if ti.dtype == torch.int16 else 77); tm.zero_()      # the caller's buffers are reused at once
            torch.cuda.synchronize()
        d, m = mb.blend()
        assert np.array_equal(m.cpu().numpy(), om), cycle
        assert np.array_equal(d.cpu().numpy(), od), cycle


def test_feather_deferred_copy_mode(gpu, oracle):
    import torch
    rng = np.random.default_rng(92)
    corners = [(0, 0), (48, 3)]
    sizes = [(80, 60), (75, 58)]
    tiles = _tiles(rng, sizes)
    res = []
    for mode in (False, "copy"):
        fb = gpu.FeatherBlender(False, 0.1)
        fb.set_deferred_level0(mode)
        fb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            fb.feed(ti, tm, c)
            if mode:
                ti.fill_(321); tm.zero_()
                torch.cuda.synchronize()
        d, m = fb.blend()
        res.append((d.cpu().numpy(), m.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("prec", [0, 1, 2])
def test_pair_stitcher_cycles_agree(gpu, prec):
    """PairStitcher's three blender cycles - eager (OpenCV's contract), deferred with private copies (the same contract),
    deferred on the caller's buffers - give identical mosaics (pitched device buffers: the 16-byte path of the copy)."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 1000, 600, 800.0
    K, Rs = synth.camera_pair(W, H, F)
    dev = torch.device("cuda:0")
    imgs = [torch.from_numpy(synth.make_tile(H, W, 5 + i)).to(dev) for i in range(2)]
    outs = []
    for deferred in (False, "copy", True):
        ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 4, prec, 0, None, "int16", deferred=deferred)
        for _ in range(2):
            out, om = ps.step()
        ps.check_plan()
        outs.append((out.cpu().numpy().copy(), om.cpu().numpy().copy()))
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1])


@pytest.mark.parametrize("prec_name", ["i16", "f32", "f16acc32"])
@pytest.mark.parametrize("off_img,off_mask,pad", [(1, 3, 5), (2, 1, 0), (3, 2, 7), (0, 0, 1)])
def test_deferred_blend_of_misaligned_device_views(gpu, oracle, prec_name, off_img, off_mask, pad):
    """CV_8UC3 tiles and masks that are views into larger device buffers: data pointers off by 1-3 bytes from a dword, row pitches
    that are no multiple of 4.  The level-0 kernels read them through aligned 12-byte windows (and, in the last collapse step, a
    wave-level test of those windows): every offset / pitch combination must give the oracle's mosaic."""
    import torch
    prec = {"i16": gpu.PREC_I16, "f32": gpu.PREC_F32, "f16acc32": gpu.PREC_F16ACC32}[prec_name]
    oprec = {"i16": oracle.I16, "f32": oracle.F32, "f16acc32": oracle.F16ACC32}[prec_name]
    rng = np.random.default_rng(1000 + off_img * 16 + off_mask * 4 + pad)
    dev = torch.device("cuda:0")
    sizes = [(301, 150), (277, 163)]
    corners = [(-13, 4), (170, -6)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for w, h in sizes]
    masks = [(rng.random((h, w)) > 0.1).astype(np.uint8) * 255 for w, h in sizes]

    def view(a, off):
        h, w = a.shape[:2]
        row = w * (a.shape[2] if a.ndim == 3 else 1)
        pitch = row + pad
        buf = torch.zeros((off + h * pitch + 16,), dtype=torch.uint8, device=dev)
        v = buf[off:off + h * pitch].view(h, pitch)[:, :row]
        v.copy_(torch.from_numpy(a.reshape(h, row)).to(dev))
        return v.view(h, w, 3) if a.ndim == 3 else v, buf

    mb = gpu.MultiBandBlender(False, 4, prec)
    mb.set_deferred_level0(True)
    ob = oracle.MultiBand(4, oprec)
    mb.prepare(corners, sizes); ob.prepare(corners, sizes)
    keep = []
    for im, m, c in zip(imgs, masks, corners):
        vi, bi = view(im, off_img)
        vm, bm = view(m, off_mask)
        assert vi.data_ptr() % 4 == off_img % 4 or off_img == 0
        keep += [vi, vm, bi, bm]
        mb.feed_u8(vi, vm, c)
        ob.feed(im.astype(np.int16), m, c)
    out_f32 = prec_name != "i16"
    d, dm = mb.blend(out_f32=out_f32)
    od, om = ob.blend(out_f32)
    assert np.array_equal(dm.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)


@pytest.mark.parametrize("deferred", [False, True])
def test_feed_cv8uc3_takes_opencvs_8bit_branch(gpu, oracle, deferred):
    """MultiBandBlender::feed accepts CV_8UC3 (createLaplacePyr's 8-bit branch, OpenCV 3.4.2 blenders.cpp): its numbers are those of the
    CV_16S branch on the converted image - isx_blender_feed(CV_8UC3) == isx_blender_feed_u8 == the oracle fed convertTo(CV_16S)."""
    rng = np.random.default_rng(77)
    tiles, corners = [], [(0, 0), (70, 6), (150, -4)]
    for i in range(3):
        h, w = 90 + 7 * i, 120 + 5 * i
        tiles.append((rng.integers(0, 256, (h, w, 3)).astype(np.uint8), np.where(rng.random((h, w)) < 0.8, 255, 0).astype(np.uint8)))
    sizes = [(t[0].shape[1], t[0].shape[0]) for t in tiles]
    for prec in (I16, gpu.PREC_F32):
        ob = oracle.MultiBand(4, prec)
        ob.prepare(corners, sizes)
        outs = []
        for entry in ("feed", "feed_u8"):
            mb = gpu.MultiBandBlender(False, 4, prec)
            mb.set_deferred_level0(deferred)
            mb.prepare(corners, sizes)
            for (img, mask), c in zip(tiles, corners):
                getattr(mb, entry)(img, mask, c)
            outs.append(mb.blend())
        for (img, mask), c in zip(tiles, corners):
            ob.feed(img.astype(np.int16), mask, c)
        od, om = ob.blend(False)
        for d, m in outs:
            assert np.array_equal(d, od) and np.array_equal(m, om), (prec, deferred)
