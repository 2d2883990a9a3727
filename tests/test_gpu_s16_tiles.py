"""GPU parity of the path a caller of cv::detail::Blender actually takes: feed() receives CV_16SC3 tiles (W:294,302).

Round 4: the last collapse step (k_collapse_roll) and the level-0 pyrDown (k_pyr_down0) take CV_16SC3 tiles too; these tests feed
shorts of the WHOLE int16 range (the guards OpenCV's int16 arithmetic carries - saturating subtract, truncating cast, wrapping
accumulate - then act), weights that are not just 0 / 1, tiles on each other's rims, 3 and 5 tiles over one place, the batched
chain, and BASELINE config 2 at full size with a CV_16SC3 feed.  Plus Blender::NO (W:276) and the conversions of W:261 / W:294.
"""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu

I16, F32, F16 = 0, 1, 2


def _s16_tiles(rng, sizes, lo=-32768, hi=32767):
    out = []
    for (w, h) in sizes:
        img = rng.integers(lo, hi + 1, (h, w, 3), dtype=np.int64).astype(np.int16)
        # a few extremes next to each other: the largest Laplacians a short can produce
        img[::7, ::5] = 32767
        img[3::11, 2::9] = -32768
        mask = rng.integers(0, 256, (h, w), dtype=np.int64).astype(np.uint8)
        mask[rng.random((h, w)) < 0.3] = 0
        mask[rng.random((h, w)) < 0.3] = 255
        out.append((img, mask))
    return out


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("layout", ["pair", "three_deep", "five_deep", "ragged"])
def test_deferred_cycle_on_full_range_cv16sc3_tiles(gpu, oracle, prec, layout):
    """Deferred cycle (references and private copies) and eager cycle = the oracle, bit for bit, for shorts of the whole range."""
    import torch
    rng = np.random.default_rng({"pair": 5, "three_deep": 6, "five_deep": 7, "ragged": 8}[layout])
    if layout == "pair":
        corners, sizes = [(-40, 7), (233, -12)], [(411, 300), (397, 290)]
    elif layout == "three_deep":        # every tile overlaps its second neighbour: three tiles over one strip (the MAXT = 3 kernel)
        corners, sizes = [(0, 0), (150, 9), (290, -6), (430, 4)], [(420, 260), (420, 255), (420, 262), (300, 250)]
    elif layout == "five_deep":         # more than three: the gathering kernel takes the last step
        corners, sizes = [(40 * i, 3 * (i % 3)) for i in range(6)], [(330, 200 + 5 * i) for i in range(6)]
    else:                               # tiny and odd sizes: rims everywhere, a tile narrower than a strip
        corners, sizes = [(3, 1), (57, -2), (9, 77)], [(71, 93), (5, 140), (131, 33)]
    tiles = _s16_tiles(rng, sizes)
    ob = oracle.MultiBand(5, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    f32 = prec != I16
    od, om = ob.blend(f32)
    for mode in (False, True, "copy"):
        mb = gpu.MultiBandBlender(False, 5, prec)
        mb.set_deferred_level0(mode)
        mb.prepare(corners, sizes)
        keep = []
        for (img, mask), c in zip(tiles, corners):
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            keep.append((ti, tm))
            mb.feed(ti, tm, c)
            if mode == "copy":          # OpenCV's contract: the caller may destroy what it fed (W:305-308)
                ti.fill_(-7), tm.fill_(99)
        d, m = mb.blend(out_f32=f32)
        assert np.array_equal(_np(m), om), (layout, mode)
        assert np.array_equal(_np(d), od), (layout, mode, np.argwhere(_np(d) != od)[:4])


@pytest.mark.parametrize("prec", [I16, F32])
def test_cv16sc3_tiles_equal_cv8uc3_tiles_when_the_values_are_bytes(gpu, prec):
    """convertTo(CV_16S) then feed == feed of the 8-bit tile (createLaplacePyr's two branches), on the fast kernels of both types,
    with pitched, misaligned device views (a 2-byte aligned CV_16SC3 row start is all the 12-byte windows may assume)."""
    import torch
    rng = np.random.default_rng(31)
    corners, sizes = [(0, 0), (300, 14)], [(500, 333), (470, 340)]
    outs = []
    for kind in ("u8", "s16"):
        mb = gpu.MultiBandBlender(False, 5, prec)
        mb.set_deferred_level0(True)
        mb.prepare(corners, sizes)
        keep = []
        r2 = np.random.default_rng(32)
        for (w, h), c in zip(sizes, corners):
            img = r2.integers(0, 256, (h, w, 3), dtype=np.int64).astype(np.uint8)
            mask = (r2.random((h, w)) > 0.2).astype(np.uint8) * 255
            if kind == "u8":
                ti = torch.from_numpy(img).cuda()
            else:       # a view that starts 2 bytes into an allocation, rows 6 w + 10 bytes apart
                pitch = w * 3 + 5
                buf = torch.zeros((h * pitch + 1,), dtype=torch.int16, device="cuda")
                ti = buf[1:].as_strided((h, w, 3), (pitch, 3, 1))
                ti.copy_(torch.from_numpy(img.astype(np.int16)).cuda())
            tm = torch.from_numpy(mask).cuda()
            keep.append((ti, tm))
            mb.feed(ti, tm, c)
        d, m = mb.blend(out_f32=prec != I16)
        outs.append((_np(d), _np(m)))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0])
    del rng


def test_full_size_4k_pair_fed_as_cv16sc3(gpu, oracle):
    """BASELINE config 2 at full size the way the reference feeds it: warp -> convertTo(CV_16S) (W:294) -> feed(CV_16SC3) (W:302) in the
    deferred cycle with private copies (the cv adapter's mode), I16 (OpenCV's arithmetic) and F32, against the oracle."""
    import torch
    W, H, F = 3840, 2160, 3000.0
    K, Rs = synth.camera_pair(W, H, F)
    imgs = [synth.make_tile(H, W, i) for i in range(2)]
    dev = torch.device("cuda:0")
    warper = gpu.CylindricalWarper().create(F)
    corners, warped16, o_warp, wmasks = [], [], [], []
    for i in range(2):
        c, wi = warper.warp(torch.from_numpy(imgs[i]).to(dev), K, Rs[i], gpu.INTER_LINEAR, gpu.BORDER_REFLECT)           # W:229
        _, wm = warper.warp(torch.full((H, W), 255, dtype=torch.uint8, device=dev), K, Rs[i], gpu.INTER_NEAREST, gpu.BORDER_CONSTANT)   # W:232
        oc, owi, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], imgs[i], 1, 2)
        _, owm, _ = oracle.warp_u8(oracle.CYL, F, K, Rs[i], np.full((H, W), 255, np.uint8), 0, 0)
        assert c == oc
        assert np.array_equal(wi.cpu().numpy(), owi), "4K warp (image-only tile kernel) differs"
        assert np.array_equal(wm.cpu().numpy(), owm), "4K mask warp (NEAREST tile kernel) differs"
        corners.append(c); o_warp.append(owi); wmasks.append(owm)
        warped16.append(gpu.convert_to(wi, np.int16))                                                                   # W:294
        assert np.array_equal(warped16[-1].cpu().numpy(), owi.astype(np.int16))
    seam = synth.seam_masks(corners, wmasks)
    sizes = [(m.shape[1], m.shape[0]) for m in wmasks]
    for prec in (gpu.PREC_I16, gpu.PREC_F32):
        f32 = prec == gpu.PREC_F32
        mb = gpu.MultiBandBlender(False, 5, prec)
        mb.set_deferred_level0("copy")
        ob = oracle.MultiBand(5, prec)
        mb.prepare(corners, sizes)
        ob.prepare(corners, sizes)
        for i in range(2):
            t16 = warped16[i].clone()
            tm = torch.from_numpy(seam[i]).to(dev)
            mb.feed(t16, tm, corners[i])
            t16.fill_(0), tm.fill_(0)       # feed() consumed them
            ob.feed(o_warp[i].astype(np.int16), seam[i], corners[i])
        d, m = mb.blend(out_f32=f32)
        od, om = ob.blend(f32)
        assert np.array_equal(m.cpu().numpy(), om)
        assert np.array_equal(d.cpu().numpy(), od), "4K blend of CV_16SC3 tiles differs (precision %d)" % prec
    torch.cuda.empty_cache()


@pytest.mark.parametrize("prec", [I16, F32, F16])
def test_batched_chain_on_cv16sc3_tiles(gpu, oracle, prec):
    """isx_blender_blend_batch with CV_16SC3 tiles: every mosaic equal to the oracle's."""
    import torch
    from imagestitch_amd.blender import blend_batch
    rng = np.random.default_rng(90)
    rigs = [([(0, 0), (200, 5)], [(330, 210), (300, 220)]), ([(10, -4), (180, 0)], [(280, 190), (310, 200)]), ([(0, 3), (260, 0)], [(400, 260), (380, 250)])]
    blenders, dsts, masks, want, keep = [], [], [], [], []
    for corners, sizes in rigs:
        tiles = _s16_tiles(rng, sizes, -3000, 3000)
        ob = oracle.MultiBand(4, prec)
        ob.prepare(corners, sizes)
        mb = gpu.MultiBandBlender(False, 4, prec)
        mb.set_deferred_level0(True)
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            ob.feed(img, mask, c)
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            keep.append((ti, tm))
            mb.feed(ti, tm, c)
        f32 = prec != I16
        want.append(ob.blend(f32))
        w, h = mb.result_size()
        blenders.append(mb)
        dsts.append(torch.empty((h, w, 3), dtype=torch.float32 if f32 else torch.int16, device="cuda"))
        masks.append(torch.empty((h, w), dtype=torch.uint8, device="cuda"))
    blend_batch(blenders, dsts, masks)
    for (od, om), d, m in zip(want, dsts, masks):
        assert np.array_equal(m.cpu().numpy(), om)
        assert np.array_equal(d.cpu().numpy(), od)


def test_blender_no_is_the_base_class(gpu, oracle):
    """Blender::createDefault(Blender::NO, false) (W:276): prepare / feed / blend of cv::detail::Blender itself, host and device mats,
    CV_16SC3 tiles (and CV_8UC3 through feed_u8), overlapping tiles (the later feed wins under its mask), masks that are not 0 / 255."""
    import torch
    rng = np.random.default_rng(4)
    corners, sizes = [(-5, 2), (60, -3), (20, 40)], [(100, 80), (90, 70), (77, 50)]
    tiles = _s16_tiles(rng, sizes)
    ob = oracle.NoBlend()
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    od, om = ob.blend()
    assert (om == 0).any() and (om != 0).any()
    for dev in (False, True):
        nb = gpu.Blender.createDefault(gpu.Blender.NO, False)
        nb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            if dev:
                nb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), c)
            else:
                nb.feed(img, mask, c)
        d, m = nb.blend()
        assert np.array_equal(_np(m), om) and np.array_equal(_np(d), od)
        with pytest.raises(gpu.IsxError) as e:
            nb.blend()
        assert e.value.code == 3     # blend() released dst_
    # CV_8UC3 through feed_u8 == the converted tile through feed
    nb = gpu.NoBlender()
    nb.prepare(corners[:2], sizes[:2])
    ob.prepare(corners[:2], sizes[:2])
    for (w, h), c in zip(sizes[:2], corners[:2]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.int64).astype(np.uint8)
        mask = (rng.random((h, w)) > 0.5).astype(np.uint8) * 255
        nb.feed_u8(img, mask, c)
        ob.feed(img.astype(np.int16), mask, c)
    d, m = nb.blend()
    od, om = ob.blend()
    assert np.array_equal(m, om) and np.array_equal(d, od)
    with pytest.raises(gpu.IsxError) as e:
        gpu.Blender.createDefault(7)
    assert e.value.code == 1


def test_convert_to_is_mat_convertTo(gpu, oracle):
    """isx_convert_to = Mat::convertTo(alpha 1, beta 0): W:261 (CV_8U -> CV_32F), W:294 (CV_32F -> CV_16S: saturate_cast of cvRound, ties
    to even, NaN / overflow like cvtss2si), their composition, and the narrowing ones - dense and pitched, host and device."""
    import torch
    rng = np.random.default_rng(12)
    u8 = rng.integers(0, 256, (37, 53, 3), dtype=np.int64).astype(np.uint8)
    assert np.array_equal(gpu.convert_to(u8, np.float32), u8.astype(np.float32))
    assert np.array_equal(gpu.convert_to(u8, np.int16), u8.astype(np.int16))
    f = (rng.standard_normal((37, 53, 3)) * 20000).astype(np.float32)
    f[0, :8, 0] = [0.5, 1.5, 2.5, -0.5, -1.5, 32767.5, -32768.5, 1e20]
    f[1, :4, 1] = [np.nan, np.inf, -np.inf, -1e20]
    assert np.array_equal(gpu.convert_to(f, np.int16), oracle.convert_f32(f, np.int16))
    assert np.array_equal(gpu.convert_to(f / 100.0, np.uint8), oracle.convert_f32(f / 100.0, np.uint8))
    s16 = rng.integers(-400, 700, (37, 53, 3), dtype=np.int64).astype(np.int16)
    assert np.array_equal(gpu.convert_to(s16, np.uint8), np.clip(s16, 0, 255).astype(np.uint8))
    assert np.array_equal(gpu.convert_to(s16, np.float32), s16.astype(np.float32))
    # device mats, the vector path (aligned, pitched) and a misaligned view
    t = torch.from_numpy(u8).cuda()
    assert np.array_equal(gpu.convert_to(t, np.int16).cpu().numpy(), u8.astype(np.int16))
    big = rng.integers(0, 256, (64, 256, 3), dtype=np.int64).astype(np.uint8)
    tb = torch.from_numpy(big).cuda()
    assert np.array_equal(gpu.convert_to(tb, np.int16).cpu().numpy(), big.astype(np.int16))
    buf = torch.zeros((64 * 800 + 3,), dtype=torch.uint8, device="cuda")
    view = buf[3:].as_strided((64, 256, 3), (800, 3, 1))
    view.copy_(tb)
    assert np.array_equal(gpu.convert_to(view, np.float32).cpu().numpy(), big.astype(np.float32))
    fm = torch.from_numpy(f).cuda()
    assert np.array_equal(gpu.convert_to(fm, np.int16).cpu().numpy(), oracle.convert_f32(f, np.int16))
    with pytest.raises(gpu.IsxError) as e:
        gpu.convert_to(u8, np.uint8)
    assert e.value.code == 2


@pytest.mark.parametrize("kind", ["cylindrical", "spherical"])
def test_the_two_warps_of_a_tile_as_calls_of_their_own(gpu, oracle, kind):
    """W:229 warp(img, LINEAR, REFLECT) and W:232 warp(mask, NEAREST, CONSTANT) through RotationWarper::warp - the tile kernels without
    the fusion - against the oracle: cameras that look past the image (z <= 0), a mask with holes (it IS read), dense (unaligned) and
    pitched destinations, sources of odd sizes."""
    import torch
    O = oracle
    pk = O.CYL if kind == "cylindrical" else O.SPH
    rng = np.random.default_rng(3)
    cases = [(640, 360, 500.0, 0.36), (333, 207, 260.0, 1.1), (97, 61, 40.0, 0.2), (1280, 720, 900.0, 0.0)]
    for (W, H, F, yaw) in cases:
        K, Rs = synth.camera_pair(W, H, F, yaw=yaw, pitch=0.07, roll=0.11)
        warper = (gpu.CylindricalWarper if kind == "cylindrical" else gpu.SphericalWarper)().create(F)
        img = rng.integers(0, 256, (H, W, 3), dtype=np.int64).astype(np.uint8)
        mask = (rng.random((H, W)) > 0.1).astype(np.uint8) * rng.integers(1, 256, (H, W), dtype=np.int64).astype(np.uint8)
        for R in Rs:
            oc, owi, roi = O.warp_u8(pk, F, K, R, img, O.LINEAR, O.BORDER_REFLECT)
            _, owm, _ = O.warp_u8(pk, F, K, R, mask, O.NEAREST, O.BORDER_CONSTANT)
            for dev in (False, True):
                si = torch.from_numpy(img).cuda() if dev else img
                sm = torch.from_numpy(mask).cuda() if dev else mask
                c, wi = warper.warp(si, K, R, gpu.INTER_LINEAR, gpu.BORDER_REFLECT)
                c2, wm = warper.warp(sm, K, R, gpu.INTER_NEAREST, gpu.BORDER_CONSTANT)
                assert c == oc and c2 == oc
                assert np.array_equal(_np(wi), owi), (kind, W, H, dev)
                assert np.array_equal(_np(wm), owm), (kind, W, H, dev)
            # pitched device destinations (dword stores)
            h, w = owm.shape
            pitch = (w * 3 + 63) // 64 * 64
            di = torch.empty((h * pitch,), dtype=torch.uint8, device="cuda").as_strided((h, w, 3), (pitch, 3, 1))
            pm = (w + 63) // 64 * 64
            dm = torch.empty((h * pm,), dtype=torch.uint8, device="cuda").as_strided((h, w), (pm, 1))
            warper.warp(torch.from_numpy(img).cuda(), K, R, gpu.INTER_LINEAR, gpu.BORDER_REFLECT, dst=di)
            warper.warp(torch.from_numpy(mask).cuda(), K, R, gpu.INTER_NEAREST, gpu.BORDER_CONSTANT, dst=dm)
            assert np.array_equal(di.cpu().numpy(), owi) and np.array_equal(dm.cpu().numpy(), owm)


def test_no_blender_refuses_a_float_result_and_a_window(gpu):
    """Blender::NO hands out its CV_16SC3 canvas: a CV_32FC3 result (whatever precision the handle was created with) and a column window
    are refused, not mis-served."""
    import torch
    from imagestitch_amd import IsxError, _lib as L
    b = gpu.NoBlender()
    img = torch.zeros((64, 80, 3), dtype=torch.int16, device="cuda")
    msk = torch.full((64, 80), 255, dtype=torch.uint8, device="cuda")
    b.prepare([(0, 0)], [(80, 64)])
    b.feed(img, msk, (0, 0))
    with pytest.raises(IsxError):
        b.blend(out_f32=True)
    b.prepare([(0, 0)], [(80, 64)])
    b.feed(img, msk, (0, 0))
    b.set_window(0, 128)
    with pytest.raises(IsxError):
        b.blend()


def test_int16_arithmetic_on_floats_equals_the_integer_forms_at_4k(gpu):
    """ISX_PREC_I16 in k_collapse_roll / k_pyr_down0 runs on integer-valued floats (round 4); the integer forms it replaced are still in the
    library (k_collapse_gather as the last step: ISX_ROLL=0, k_pyr_down_multi at level 0: ISX_PD0=0).  CV_16SC3 tiles of the whole short
    range with saturating differences and masks with holes, 4096 x 2160, 7 bands: same mosaic (tools/probes/int16_forms_probe.py holds
    the larger sizes, profiles/round4_int16_forms.json)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = os.path.join(root, "tools", "probes", "int16_forms_probe.py")
    got = {}
    for roll, pd0 in (("1", "1"), ("0", "0")):
        env = dict(os.environ, ISX_ROLL=roll, ISX_PD0=pd0)
        out = subprocess.check_output([sys.executable, probe, "--child", "s16", "4096", "2160", "7"], env=env, text=True)
        line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
        got[(roll, pd0)] = line[1]
        assert ("collapse_roll" if roll == "1" else "collapse_gather") in " ".join(line[2:])
    assert got[("1", "1")] == got[("0", "0")]


@pytest.mark.parametrize("prec", ["i16", "f32"])
def test_dense_result_of_odd_width_equals_the_pitched_one_and_the_oracle(gpu, oracle, prec):
    """blend() into a dense CV_16SC3 cv::Mat whose width is odd (rows start on 2-byte boundaries: what dst.create() gives the reference's
    result, W:313) takes k_collapse_roll's vector stores since round 4 (they are typed 2-byte aligned); same bytes as a result with 4-byte
    aligned rows, an offset view included, and as the oracle's."""
    import torch
    P = {"i16": (gpu.PREC_I16, oracle.I16), "f32": (gpu.PREC_F32, oracle.F32)}[prec]
    h, w = 230, 301
    tiles = [synth.make_tile(h, w, 60 + i) for i in range(2)]
    corners = [(0, 0), (188, 7)]                      # mosaic 489 x 237: an odd width
    sizes = [(w, h), (w, h)]
    masks = [np.full((h, w), 255, np.uint8) for _ in range(2)]
    masks[0][:, 250:] = 0; masks[1][:, :40] = 0
    ob = oracle.MultiBand(4, P[1])
    ob.prepare(corners, sizes)
    for i in range(2):
        ob.feed(tiles[i].astype(np.int16), masks[i], corners[i])
    od, om = ob.blend(False)
    fh, fw = od.shape[:2]
    assert fw % 2 == 1
    outs = []
    for layout in ("dense", "pitched", "offset"):
        b = gpu.MultiBandBlender(False, 4, P[0], 0)
        b.set_deferred_level0(True)
        b.prepare(corners, sizes)
        keep = []
        for i in range(2):
            t16 = torch.from_numpy(tiles[i].astype(np.int16)).cuda(); m = torch.from_numpy(masks[i]).cuda()
            keep += [t16, m]
            b.feed(t16, m, corners[i])
        if layout == "dense":
            dst = torch.empty((fh, fw, 3), dtype=torch.int16, device="cuda"); dm = torch.empty((fh, fw), dtype=torch.uint8, device="cuda")
            assert (dst.stride(0) * 2) % 4 == 2
        elif layout == "pitched":
            dst = torch.empty((fh, fw + 1, 3), dtype=torch.int16, device="cuda")[:, :fw]; dm = torch.empty((fh, fw + 3), dtype=torch.uint8, device="cuda")[:, :fw]
        else:                                          # rows 4-byte aligned in pitch but starting 2 bytes into a dword
            dst = torch.empty((fh, fw + 3, 3), dtype=torch.int16, device="cuda")[:, 1:fw + 1]; dm = torch.empty((fh, fw + 5), dtype=torch.uint8, device="cuda")[:, 1:fw + 1]
        b.blend(dst, dm)
        assert b.last_path()["last_step"] == "collapse_roll"
        outs.append((dst.cpu().numpy().copy(), dm.cpu().numpy().copy()))
    for d, m in outs:
        assert np.array_equal(d, od) and np.array_equal(m, om)
