"""Known-answer and property tests of the oracle + bit-exact agreement of the two
independent restatements (oracle/oracle.c vs oracle/oracle_np.py)."""
import numpy as np
import pytest

from oracle import oracle_np as N


def test_scalar_conversions(oracle):
    L = oracle.lib()
    assert [L.orc_cvround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999)] == [0, 2, 2, 0, -2, 2]   # round-half-even
    assert L.orc_cvround(float("nan")) == -2 ** 31 and L.orc_cvround(3e9) == -2 ** 31 and L.orc_cvround(-3e9) == -2 ** 31
    assert [L.orc_f2i_trunc(v) for v in (-543.0966, 542.99, -0.9)] == [-543, 542, 0]                 # W:83-86
    for p, n, exp101, exp in ((-1, 5, 1, 0), (-2, 5, 2, 1), (5, 5, 3, 4), (6, 5, 2, 3), (-7, 3, 1, 0)):
        assert L.orc_border_interpolate(p, n, 4) == exp101 and L.orc_border_interpolate(p, n, 2) == exp
    assert L.orc_border_interpolate(-3, 1, 2) == 0 and L.orc_border_interpolate(9, 7, 0) == -1
    # f16 rounding incl. ties, subnormals, overflow
    vals = np.array([1.0, 1.00048828125, 1.0009765625 + 2 ** -12, 65519.9, 65520.0, 6e-8, 2.98e-8, 3e-8, -2049.0, 0.1], np.float32)
    assert np.array_equal(oracle.f16_round(vals), vals.astype(np.float16).astype(np.float32))
    # weight of a set mask pixel is exactly 1.0f: 255 * (float)(1/255.)
    assert np.float32(255) * np.float32(1. / 255.) == np.float32(1.0)


def test_remap_known_answers(oracle):
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:9, 0:11].astype(np.float32)
    assert np.array_equal(oracle.remap(src, xx, yy, 1, 2), src)                      # identity
    sh = oracle.remap(src, xx - 2, yy + 1, 1, 2)                                      # integer shift + BORDER_REFLECT
    assert np.array_equal(sh[0, 2:], src[1, :9]) and np.array_equal(sh[:, 1], oracle.remap(src, xx - 2, yy + 1, 0, 2)[:, 1])
    assert np.array_equal(sh[0, 1], src[1, 0]) and np.array_equal(sh[0, 0], src[1, 1])   # fedcba|abc: -1 -> 0, -2 -> 1
    assert np.array_equal(sh[8, 5], src[8, 3])                                        # row 9 reflects to row 8
    half = oracle.remap(src, xx + 0.5, yy, 1, 2)                                      # half pixel: (a*16384 + b*16384 + 16384) >> 15
    a, b = src[:, :-1].astype(int), src[:, 1:].astype(int)
    assert np.array_equal(half[:, :-1], ((a * 16384 + b * 16384 + 16384) >> 15).astype(np.uint8))
    # (-1,-1) sentinel of mapBackward (W:61) with BORDER_REFLECT samples pixel (0,0): no black fill
    m1 = np.full((2, 2), -1, np.float32)
    assert np.array_equal(oracle.remap(src, m1, m1, 1, 2)[0, 0], src[0, 0])
    # masks: NEAREST + CONSTANT of an all-255 mask is {0,255} only, 0 outside
    mask = np.full((9, 11), 255, np.uint8)
    out = oracle.remap(mask, xx * 1.7 - 3, yy * 1.7 - 3, 0, 0)
    assert set(np.unique(out)) == {0, 255} and out[0, 0] == 0
    # rounding of .5 in NEAREST is half-even: x = 0.5 -> 0, 1.5 -> 2
    line = np.arange(11, dtype=np.uint8)[None, :].repeat(3, 0)
    near = oracle.remap(line, np.array([[0.5, 1.5, 2.5]], np.float32), np.zeros((1, 3), np.float32), 0, 0)
    assert near.tolist() == [[0, 2, 2]]


def test_pyramid_known_answers(oracle):
    const = np.full((16, 24, 3), 1234, np.int16)
    assert np.all(oracle.pyr_down(const) == 1234) and np.all(oracle.pyr_up(const) == 1234)    # weights sum to 256 / 64
    cf = np.full((8, 8), 0.3, np.float32)
    assert np.allclose(oracle.pyr_down(cf), 0.3, atol=1e-6) and np.allclose(oracle.pyr_up(cf), 0.3, atol=1e-6)
    imp = np.zeros((17, 17), np.float32)
    imp[8, 8] = 256.0
    # impulse response of [1 4 6 4 1]^2 / 256 seen through the stride-2 decimation: taps 1 6 1 around the centre
    assert np.array_equal(oracle.pyr_down(imp)[3:6, 3:6], np.outer([1, 6, 1], [1, 6, 1]).astype(np.float32))
    assert oracle.pyr_down(imp).sum() == 64.0
    upi = oracle.pyr_up(np.pad(np.array([[64.0]], np.float32), 2))     # pyrUp impulse: [1 4 6 4 1]^2 / 64
    assert np.array_equal(upi[2:7, 2:7], np.outer([1, 4, 6, 4, 1], [1, 4, 6, 4, 1]).astype(np.float32))
    row = np.array([[10, 20, 40, 80]], np.int16)                                      # REFLECT_101 at both edges
    d = oracle.pyr_down(row)
    # horizontal taps of x=0: 6*10 + 4*(20+20) + 40 + 40 = 300 ; vertical (1 row): 16 * 300 -> (4800+128)>>8 = 19
    assert d[0, 0] == (16 * 300 + 128) >> 8
    u = oracle.pyr_up(np.array([[64, 128]], np.int16))                                # pyrUp edge formulas
    # row pass: [6*64+2*128, 4*(64+128), 64+7*128, 8*128] = [640, 768, 960, 1024]; 1 source row -> rows x8
    assert u[0].tolist() == [(8 * v + 32) >> 6 for v in (640, 768, 960, 1024)]
    one = oracle.pyr_up(np.array([[5]], np.int16))
    assert one.tolist() == [[5, 5], [5, 5]]                                           # n == 1: s*8 both ways


@pytest.mark.parametrize("shape", [(16, 16, 3), (17, 23, 3), (8, 6), (2, 2, 3), (1, 5, 3), (4, 1, 3), (33, 64, 1)])
def test_c_and_numpy_pyramids_agree(oracle, shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(-32768, 32768, shape).astype(np.int16)
    f = (rng.standard_normal(shape) * 100).astype(np.float32)
    for arr in (a, f):
        assert np.array_equal(oracle.pyr_down(arr), N.pyr_down(arr))
        assert np.array_equal(oracle.pyr_up(arr), N.pyr_up(arr))


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (5, 3), (2, 9)])
def test_fp32_level1_of_bytes_is_a_multiple_of_one_256th(oracle, shape):
    """What the product's Q8 records lean on (blend.hip load_px_planar): in fp32, pyrDown of integers 0..255 is k / 256 with an integer
    k <= 255 * 256 - exactly, in both restatements - so three unsigned shorts hold a level-1 pixel and (float)k * 2^-8 gives it back bit for bit.
    (Level 2 is NOT: it needs 24 bits per channel; the extremes - all 255, alternating 0 / 255 - are in the set.)"""
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    h, w = shape
    imgs = [rng.integers(0, 256, (h, w, 3)), np.full((h, w, 3), 255), (np.indices((h, w)).sum(0) % 2 * 255)[..., None].repeat(3, 2),
            rng.integers(0, 2, (h, w, 3)) * 255]
    not_q8 = 0
    for img in imgs:
        a = img.astype(np.float32)
        for down in (oracle.pyr_down, N.pyr_down):
            g1 = down(a)
            k = g1 * np.float32(256)
            assert np.array_equal(k, np.rint(k)) and k.min() >= 0 and k.max() <= 255 * 256, (shape, down.__module__)
            back = k.astype(np.uint16).astype(np.float32) * np.float32(1 / 256)
            assert np.array_equal(back.view(np.uint32), g1.view(np.uint32))
            g2 = down(g1) * np.float32(256)
            not_q8 += int(not np.array_equal(g2, np.rint(g2)))
    assert not_q8 > 0 or h * w < 200          # (the argument stops at level 1; tiny images can fall on the grid by accident)


def test_c_and_numpy_remap_agree(oracle):
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    xm = (rng.random((40, 60)) * 70 - 8).astype(np.float32)
    ym = (rng.random((40, 60)) * 50 - 6).astype(np.float32)
    xm[0, 0] = ym[0, 0] = -1
    xm[1, 1] = np.nan
    xm[2, 2] = 1e12
    ym[3, 3] = -1e12
    xm[4, 4], ym[4, 4] = 52.5, 36.5
    for interp in (0, 1):
        for border in (0, 1, 2, 4):
            for s in (src, src[:, :, 0].copy(), src.astype(np.float32)):
                assert np.array_equal(oracle.remap(s, xm, ym, interp, border), N.remap(s, xm, ym, interp, border), equal_nan=True)


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("bands", [0, 1, 3, 5])
def test_c_and_numpy_multiband_agree(oracle, prec, bands):
    rng = np.random.default_rng(10 * prec + bands)
    corners, sizes = [(-5, 3), (40, -2), (20, 30)], [(70, 50), (64, 57), (33, 41)]
    mo, mn = oracle.MultiBand(bands, prec), N.MultiBand(bands, prec)
    mo.prepare(corners, sizes)
    mn.prepare(corners, sizes)
    assert mo.num_bands == mn.L
    for c, s in zip(corners, sizes):
        img = rng.integers(-300, 600, (s[1], s[0], 3)).astype(np.int16)
        mask = (rng.random((s[1], s[0])) > 0.3).astype(np.uint8) * 255
        mo.feed(img, mask, c)
        mn.feed(img, mask, c)
    for i in range(mo.num_bands + 1):
        lap, w = mo.level(i)
        assert np.array_equal(lap, mn.lap[i]) and np.array_equal(w, mn.wgt[i])
    d, m = mo.blend(prec != 0)
    d2, m2 = mn.blend(prec != 0)
    assert np.array_equal(d, d2) and np.array_equal(m, m2)


def test_multiband_properties(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (96, 128, 3)).astype(np.int16)
    full = np.full((96, 128), 255, np.uint8)
    # single image, full mask: F32 returns the image to a few ulp; I16 within the documented -1 truncation bias
    mb = oracle.MultiBand(4, oracle.F32)
    mb.prepare([(0, 0)], [(128, 96)])
    mb.feed(img, full, (0, 0))
    d, m = mb.blend(True)
    assert np.all(m == 255) and np.abs(d - img).max() < 1e-2
    mb = oracle.MultiBand(4, oracle.I16)
    mb.prepare([(0, 0)], [(128, 96)])
    mb.feed(img, full, (0, 0))
    d, m = mb.blend(False)
    assert np.abs(d.astype(int) - img).max() <= 8 and (d.astype(int) - img).mean() < 0      # biased low: 100/(1+1e-5) -> 99
    assert oracle.lib().orc_f2i_trunc(np.float32(100) / (np.float32(1) + np.float32(1e-5))) == 99
    # identical images, complementary masks -> the image again (F32)
    left = full.copy(); left[:, 64:] = 0
    right = full.copy(); right[:, :64] = 0
    mb = oracle.MultiBand(3, oracle.F32)
    mb.prepare([(0, 0), (0, 0)], [(128, 96), (128, 96)])
    mb.feed(img, left, (0, 0)); mb.feed(img, right, (0, 0))
    d, m = mb.blend(True)
    assert np.all(m == 255) and np.abs(d - img).max() < 1e-2
    # dst_mask is 255 exactly where a fed mask is set; padded sizes are multiples of 2^L; bands clamp
    mb = oracle.MultiBand(5, oracle.I16)
    mb.prepare([(3, 7)], [(100, 60)])
    hole = full[:60, :100].copy(); hole[10:20, 30:50] = 0
    mb.feed(img[:60, :100], hole, (3, 7))
    lap0, w0 = mb.level(0)
    assert lap0.shape[0] % 32 == 0 and lap0.shape[1] % 32 == 0 and lap0.shape[:2] == (64, 128)
    d, m = mb.blend(False)
    assert np.array_equal(m, hole) and np.all(d[m == 0] == 0)
    mb = oracle.MultiBand(9, oracle.I16)
    mb.prepare([(0, 0)], [(20, 9)])
    assert mb.num_bands == 5                                                           # ceil(log2(20)) = 5


def test_linear_blend_properties(oracle):
    """A13: equal constant images -> identity in the overlap; m1 + m2 = 1 wherever both are valid."""
    a = np.full((50, 80, 3), 100.0, np.float32)
    b = np.full((50, 70, 3), 100.0, np.float32)
    rc, pano, seam = oracle.blend_pair_linear(a, b, (0, 0), (45, 0))
    assert rc == 0 and pano.shape == (50, 115, 3)
    assert np.allclose(pano, 100.0, atol=1e-3)
    rc, _, _ = oracle.blend_pair_linear(a, b, (0, 0), (200, 0))
    assert rc == 1                                                                     # no overlap: B:182-183


def test_dilate_and_distance_transform_known_answers(oracle):
    m = np.zeros((9, 12), np.uint8)
    m[4, 5] = 255
    d = oracle.dilate_rect(m, 4, 3)              # anchor (2, 1): columns x-2..x+1 -> 4..7, rows y-1..y+1 -> 3..5
    exp = np.zeros_like(m)
    exp[3:6, 4:8] = 255
    assert np.array_equal(d, exp)
    d20 = oracle.dilate_rect(m, 20, 20)          # 20x20 (W:286): window [x-10, x+9] around the pixel, mirrored for the set pixel
    ys, xs = np.nonzero(d20)
    assert xs.min() == max(5 - 9, 0) and xs.max() == min(5 + 10, 11)
    rng = np.random.default_rng(0)
    mk = (rng.random((40, 55)) > 0.02).astype(np.uint8) * 255
    dt = oracle.distance_transform_l1(mk)
    ys, xs = np.nonzero(mk == 0)
    Y, X = np.mgrid[0:40, 0:55]
    brute = np.min(np.abs(Y[..., None] - ys) + np.abs(X[..., None] - xs), axis=2)
    assert np.array_equal(dt, brute.astype(np.float32))          # chamfer {1, 2} == exact city-block distance
    full = oracle.distance_transform_l1(np.full((3, 5), 255, np.uint8))
    assert full.tolist() == [[8193.0] * 5, [8193.0, 8194.0, 8194.0, 8194.0, 8193.0], [8193.0] * 5]   # INIT_DIST0 border ring
    w = oracle.feather_weight_map(np.pad(np.full((30, 30), 255, np.uint8), 1), 0.1)
    assert w[0, 0] == 0.0 and w[1, 1] == np.float32(1.0) * np.float32(0.1) and w[16, 16] == 1.0 and abs(w[5, 16] - 0.5) < 1e-6


def test_feather_properties(oracle):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (60, 80, 3)).astype(np.int16)
    full = np.full((60, 80), 255, np.uint8)
    fb = oracle.Feather(0.1)
    fb.prepare([(0, 0)], [(80, 60)])
    fb.feed(img, full, (0, 0))
    d, m = fb.blend()
    assert np.all(m == 255) and np.abs(d.astype(int) - img).max() <= 1 and (d.astype(int) - img).max() <= 0   # weight 1, -1 truncation bias


def test_gain_apply_known_answers(oracle):
    """GainCompensator::apply = saturate_cast<uchar>(cvRound((double)v * gain)) (W:241-244)."""
    v = np.arange(256, dtype=np.uint8)
    assert np.array_equal(oracle.gain_apply(v, 1.0), v)
    assert np.array_equal(oracle.gain_apply(v, 2.0), np.minimum(v.astype(int) * 2, 255))
    # ties go to even: 1 * 0.5 = 0.5 -> 0, 3 * 0.5 = 1.5 -> 2, 5 * 0.5 = 2.5 -> 2
    assert list(oracle.gain_apply(np.array([1, 3, 5, 7], np.uint8), 0.5)) == [0, 2, 2, 4]
    assert list(oracle.gain_apply(np.array([0, 1, 200], np.uint8), -1.0)) == [0, 0, 0]
    # cvRound on NaN / beyond int range is INT_MIN -> saturates to 0 (x86 cvtsd2si), not 255
    assert list(oracle.gain_apply(np.array([0, 1, 200], np.uint8), float("inf"))) == [0, 0, 0]
    assert list(oracle.gain_apply(np.array([1, 200], np.uint8), 1e8)) == [255, 0]
    g = 1.0379
    ref = np.clip(np.rint(v.astype(np.float64) * g), 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.gain_apply(v, g), ref)


def test_seam_estimate_known_answers(oracle):
    """estimateSeam S:806-957 / computeCosts S:733-803 on cases with an obvious answer."""
    h, w = 40, 30
    lab = np.full((h, w), 1, np.int32)
    a = np.zeros((h, w, 3), np.float32)
    # identical images: every step costs 0, ties keep step 1 (straight), the seam drifts only as far as it must
    seam, horiz = oracle.seam_estimate(a, a, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h), (10, 0), (10, h - 1))
    assert not horiz and len(seam) == h and (seam[:, 0] == 10).all() and (seam[:, 1] == np.arange(h)).all()
    seam, _ = oracle.seam_estimate(a, a, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h), (10, 0), (14, h - 1))
    assert tuple(seam[0]) == (10, 0) and tuple(seam[-1]) == (14, h - 1)
    assert (np.diff(seam[:, 1]) == 1).all() and (np.abs(np.diff(seam[:, 0])) <= 1).all()
    # swapped tips: same path, reported from p1 to p2
    back, _ = oracle.seam_estimate(a, a, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h), (14, h - 1), (10, 0))
    assert tuple(back[0]) == (14, h - 1) and tuple(back[-1]) == (10, 0)
    # a zero-cost valley between columns 19 and 20 (the images differ everywhere else): the seam runs along it
    b = a.copy()
    b[:, :, :] = 50.0
    b[:, 19:21, :] = 0.0
    seam, _ = oracle.seam_estimate(a, b, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h), (20, 0), (20, h - 1))
    assert (seam[:, 0] == 20).all()
    cv, ch = oracle.seam_costs(a, b, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h))
    assert cv.shape == (h, w + 1) and ch.shape == (h + 1, w)
    assert cv[5, 0] == 3 * 255.0 ** 2 and cv[5, w] == 3 * 255.0 ** 2          # badRegionCost outside the component (x == 0, x == width)
    assert cv[5, 20] == 0.0 and cv[5, 10] == 3 * 50.0 ** 2                     # (|I1(x-1)-I2(x)|^2 + |I1(x)-I2(x-1)|^2) / 2
    # horizontal seam when the tips are further apart in x than in y
    seam, horiz = oracle.seam_estimate(a, a, (0, 0), (0, 0), (0, 0), lab, 1, (0, 0, w, h), (0, 7), (w - 1, 9))
    assert horiz and len(seam) == w and tuple(seam[0]) == (0, 7) and tuple(seam[-1]) == (w - 1, 9)
    # a wall of another component between the tips: not reachable -> empty seam (`return false`)
    lab2 = lab.copy(); lab2[20, :] = 2
    seam, _ = oracle.seam_estimate(a, a, (0, 0), (0, 0), (0, 0), lab2, 1, (0, 0, w, h), (10, 0), (10, h - 1))
    assert len(seam) == 0


def test_blender_no_known_answers(oracle):
    """cv::detail::Blender (Blender::NO, W:276): feed copies under the mask and ORs the mask, blend zeroes what no mask covered."""
    nb = oracle.NoBlend()
    nb.prepare([(0, 0), (2, 1)], [(4, 3), (4, 3)])
    assert nb.result_size() == (6, 4)
    a = np.full((3, 4, 3), 100, np.int16)
    b = np.full((3, 4, 3), -7, np.int16)
    ma = np.array([[255, 255, 0, 0]] * 3, np.uint8)
    mb = np.array([[0, 4, 9, 0]] * 3, np.uint8)
    nb.feed(a, ma, (0, 0))
    nb.feed(b, mb, (2, 1))
    d, m = nb.blend()
    want_m = np.zeros((4, 6), np.uint8)
    want_m[0:3, 0:2] = 255
    want_m[1:4, 3] |= 4
    want_m[1:4, 4] |= 9
    assert np.array_equal(m, want_m)
    assert np.all(d[m == 0] == 0)
    assert np.all(d[0:3, 0:2] == 100) and np.all(d[1:4, 3:5] == -7)


def test_convert_known_answers(oracle):
    """Mat::convertTo(CV_16S / CV_8U) of floats (W:294, W:315's input): saturate_cast<T>(cvRound(v)), ties to even, cvtss2si's indefinite."""
    f = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 32767.4, 32767.5, 40000.0, -32768.5, -40000.0, 1e20, -1e20, np.nan, np.inf], np.float32)
    assert oracle.convert_f32(f, np.int16).tolist() == [0, 2, 2, 0, -2, 32767, 32767, 32767, -32768, -32768, -32768, -32768, -32768, -32768]
    g = np.array([0.5, 1.5, 254.5, 255.5, 300.0, -3.0, np.nan, 1e20], np.float32)
    assert oracle.convert_f32(g, np.uint8).tolist() == [0, 2, 254, 255, 255, 0, 0, 0]
