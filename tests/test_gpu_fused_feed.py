"""GPU parity of the fused feed of mode 2 (round 5, k_feed_pd0): feed() under OpenCV's contract (W:302-308) is ONE pass over the caller's
CV_8UC3 / CV_16SC3 device tile - level 1 of the tile's pyramid and the private copy come out of the same read - and the private copy of
a CV_16SC3 tile that holds only bytes (what convertTo(CV_16S) of a warped CV_8UC3 image produces, W:294) is kept as CV_8UC3.

What must hold: the result is the oracle's, bit for bit, whether the tiles are bytes (narrowed copies confirmed), are not (a segment
escapes, the cycle is widened before the last step), or stop being bytes in a later cycle of the same blender; the caller may destroy what
it fed as soon as feed() returns; and every layout that sends blend() down another road (windows, more than 20 tiles, level
introspection, a third tile over one place, one band) still finds level 1 and the copies in the form it expects.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

I16, F32, F16 = 0, 1, 2


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


def _tiles(rng, sizes, kind):
    """kind: 'bytes' = shorts in [0, 255]; 'one' = bytes but for ONE value per tile; 'full' = the whole short range"""
    out = []
    for (w, h) in sizes:
        if kind == "full":
            img = rng.integers(-32768, 32768, (h, w, 3), dtype=np.int64).astype(np.int16)
        else:
            img = rng.integers(0, 256, (h, w, 3), dtype=np.int64).astype(np.int16)
            if kind == "one":
                y, x, c = int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 3))
                img[y, x, c] = [256, -1, 32767, -32768][int(rng.integers(0, 4))]
        mask = rng.integers(0, 256, (h, w), dtype=np.int64).astype(np.uint8)
        mask[rng.random((h, w)) < 0.4] = 255
        mask[rng.random((h, w)) < 0.2] = 0
        out.append((img, mask))
    return out


def _oracle_blend(oracle, bands, prec, corners, sizes, tiles):
    ob = oracle.MultiBand(bands, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img, mask, c)
    return ob.blend(prec != I16)


def _gpu_blend(gpu, mb, prec, corners, sizes, tiles, poison=True, u8=False):
    import torch
    mb.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ti = torch.from_numpy(img.astype(np.uint8) if u8 else img).cuda()
        tm = torch.from_numpy(mask).cuda()
        (mb.feed_u8 if u8 else mb.feed)(ti, tm, c)
        if poison:      # OpenCV's contract: the caller may destroy what it fed (W:305-308)
            ti.fill_(77 if u8 else -7), tm.fill_(99)
    d, m = mb.blend(out_f32=prec != I16)
    return _np(d), _np(m)


LAYOUTS = {
    "pair": ([(-40, 7), (233, -12)], [(411, 300), (397, 290)]),
    "ragged": ([(3, 1), (57, -2), (9, 77)], [(71, 93), (5, 140), (131, 33)]),
    "wide": ([(0, 0), (700, 5)], [(1100, 210), (900, 200)]),            # several block columns per tile: interior and rim blocks of k_feed_pd0
    "three_deep": ([(0, 0), (150, 9), (290, -6), (430, 4)], [(420, 260), (420, 255), (420, 262), (300, 250)]),
    "five_deep": ([(40 * i, 3 * (i % 3)) for i in range(6)], [(330, 200 + 5 * i) for i in range(6)]),      # the gathering kernel takes the last step
}


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("layout", sorted(LAYOUTS))
@pytest.mark.parametrize("kind", ["bytes", "one", "full"])
def test_fused_feed_equals_the_oracle(gpu, oracle, prec, layout, kind):
    corners, sizes = LAYOUTS[layout]
    import zlib
    rng = np.random.default_rng(zlib.crc32((layout + kind).encode()))
    tiles = _tiles(rng, sizes, kind)
    od, om = _oracle_blend(oracle, 5, prec, corners, sizes, tiles)
    mb = gpu.MultiBandBlender(False, 5, prec)
    mb.set_deferred_level0("copy")
    d, m = _gpu_blend(gpu, mb, prec, corners, sizes, tiles)
    assert np.array_equal(m, om), (layout, kind)
    assert np.array_equal(d, od), (layout, kind, np.argwhere(d != od)[:4])
    path = mb.feed_path()
    assert path["fused_tiles"] == len(tiles)
    assert path["narrowed"] == ("confirmed" if kind == "bytes" else "widened"), path


@pytest.mark.parametrize("prec", [I16, F32])
def test_a_blender_that_was_violated_once_keeps_wide_copies_and_one_that_was_not_keeps_narrowing(gpu, oracle, prec):
    corners, sizes = LAYOUTS["pair"]
    rng = np.random.default_rng(3)
    mb = gpu.MultiBandBlender(False, 5, prec)
    mb.set_deferred_level0("copy")
    seen = []
    for kind in ("bytes", "bytes", "one", "bytes", "full", "bytes"):
        tiles = _tiles(rng, sizes, kind)
        od, om = _oracle_blend(oracle, 5, prec, corners, sizes, tiles)
        d, m = _gpu_blend(gpu, mb, prec, corners, sizes, tiles)
        assert np.array_equal(m, om) and np.array_equal(d, od), kind
        seen.append(mb.feed_path()["narrowed"])
    assert seen == ["confirmed", "confirmed", "widened", "none", "none", "none"], seen


@pytest.mark.parametrize("bands", [1, 2, 3, 4, 6])
def test_fused_feed_with_other_band_counts(gpu, oracle, bands):
    """1 band: the first collapse step is also the last; 3: k_collapse_top produces level 1 (no planar level 1); 2, 4, 6: as 5."""
    corners, sizes = LAYOUTS["wide"]
    rng = np.random.default_rng(40 + bands)
    for prec in (I16, F32):
        tiles = _tiles(rng, sizes, "bytes")
        od, om = _oracle_blend(oracle, bands, prec, corners, sizes, tiles)
        mb = gpu.MultiBandBlender(False, bands, prec)
        mb.set_deferred_level0("copy")
        d, m = _gpu_blend(gpu, mb, prec, corners, sizes, tiles)
        assert np.array_equal(m, om) and np.array_equal(d, od), (bands, prec)
        assert mb.feed_path() == {"fused_tiles": 2, "narrowed": "confirmed"}


def test_fused_feed_of_cv8uc3_tiles(gpu, oracle):
    """feed_u8 in mode 2: the same pass, the copy is the tile's own type (nothing to confirm)."""
    corners, sizes = LAYOUTS["wide"]
    rng = np.random.default_rng(9)
    for prec in (I16, F32, F16):
        tiles = _tiles(rng, sizes, "bytes")
        od, om = _oracle_blend(oracle, 5, prec, corners, sizes, tiles)
        mb = gpu.MultiBandBlender(False, 5, prec)
        mb.set_deferred_level0("copy")
        d, m = _gpu_blend(gpu, mb, prec, corners, sizes, tiles, u8=True)
        assert np.array_equal(m, om) and np.array_equal(d, od), prec
        assert mb.feed_path() == {"fused_tiles": 2, "narrowed": "none"}


def test_level_introspection_after_fused_feeds(gpu, oracle):
    """isx_blender_debug_level replays the recorded tiles through the eager feed: from the private copies, whose type is settled first."""
    corners, sizes = LAYOUTS["pair"]
    rng = np.random.default_rng(12)
    import torch
    for kind in ("bytes", "one"):
        tiles = _tiles(rng, sizes, kind)
        ob = oracle.MultiBand(4, I16)
        ob.prepare(corners, sizes)
        mb = gpu.MultiBandBlender(False, 4, I16)
        mb.set_deferred_level0("copy")
        mb.prepare(corners, sizes)
        for (img, mask), c in zip(tiles, corners):
            ob.feed(img, mask, c)
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            mb.feed(ti, tm, c)
            ti.fill_(-3), tm.fill_(1)
        for lvl in range(5):
            gl, gw = mb.level(lvl)
            ol, ow = ob.level(lvl)
            assert np.array_equal(gl, ol) and np.array_equal(gw, ow), (kind, lvl)
        d, m = mb.blend()
        od, om = ob.blend(False)
        assert np.array_equal(_np(m), om) and np.array_equal(_np(d), od), kind


def test_more_than_twenty_fused_tiles_and_windows(gpu, oracle):
    """Column strips (more than 20 recorded tiles) and a caller's column window blend subsets of the records: level 1 from feed() is whole,
    the strips read what they need of it."""
    rng = np.random.default_rng(77)
    n = 24
    corners = [(90 * i, int(rng.integers(-6, 7))) for i in range(n)]
    sizes = [(160, 120 + int(rng.integers(0, 9))) for _ in range(n)]
    import os
    old_tab = os.environ.get("ISX_TAB")
    try:
        for tab in ("1", "0"):      # round 5: one chain over a device-resident table of the tiles; ISX_TAB=0: round 4's column strips
            os.environ["ISX_TAB"] = tab
            for kind in ("bytes", "one"):
                tiles = _tiles(rng, sizes, kind)
                od, om = _oracle_blend(oracle, 3, I16, corners, sizes, tiles)
                mb = gpu.MultiBandBlender(False, 3, I16)
                mb.set_deferred_level0("copy")
                d, m = _gpu_blend(gpu, mb, I16, corners, sizes, tiles)
                assert mb.last_path()["cycle"] == ("deferred_table" if tab == "1" else "deferred_strips")
                assert np.array_equal(m, om) and np.array_equal(d, od), (kind, tab)
    finally:
        if old_tab is None:
            os.environ.pop("ISX_TAB", None)
        else:
            os.environ["ISX_TAB"] = old_tab
    # a caller's window over a pair
    import torch
    corners, sizes = LAYOUTS["wide"]
    tiles = _tiles(rng, sizes, "bytes")
    od, om = _oracle_blend(oracle, 5, F32, corners, sizes, tiles)
    mb = gpu.MultiBandBlender(False, 5, F32)
    mb.set_deferred_level0("copy")
    mb.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        mb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), c)
    mb.set_window(256, 896)
    d, m = mb.blend(out_f32=True)
    assert np.array_equal(_np(m), om[:, 256:896]) and np.array_equal(_np(d), od[:, 256:896])


def test_fused_feed_switched_off_gives_the_same_bits(gpu, oracle):
    """ISX_FEED_FUSE=0 / ISX_FEED_NARROW=0 (round 4's separate copy + pyrDown; CV_16SC3 copies) in child processes: hashes equal."""
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, numpy as np, torch, imagestitch_amd as I
rng = np.random.default_rng(5)
corners, sizes = [(0, 0), (700, 5)], [(1100, 210), (900, 200)]
h = hashlib.sha256()
for prec in (0, 1, 2):
    for lo, hi in ((0, 256), (-32768, 32768)):
        mb = I.MultiBandBlender(False, 5, prec); mb.set_deferred_level0("copy"); mb.prepare(corners, sizes)
        for (w, hh), c in zip(sizes, corners):
            img = rng.integers(lo, hi, (hh, w, 3), dtype=np.int64).astype(np.int16)
            mask = (rng.random((hh, w)) > 0.3).astype(np.uint8) * 255
            mb.feed(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), c)
        d, m = mb.blend(out_f32=prec != 0)
        h.update(d.cpu().numpy().tobytes()); h.update(m.cpu().numpy().tobytes())
print("HASH", h.hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hashes = []
    for env in ({}, {"ISX_FEED_FUSE": "0"}, {"ISX_FEED_NARROW": "0"}):
        e = dict(os.environ, PYTHONPATH=root, **env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        hashes.append([ln for ln in out.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1] == hashes[2], hashes


@pytest.mark.parametrize("prec", [I16, F32])
def test_narrowing_can_be_switched_off_per_blender(gpu, oracle, prec):
    """isx_blender_set_narrow_copies(0): the private copies stay CV_16SC3, blend() never polls the device - same bits; switching it
    on again narrows again; and switching it off inside a cycle that already holds narrowed tiles is refused."""
    corners, sizes = LAYOUTS["pair"]
    rng = np.random.default_rng(11)
    mb = gpu.MultiBandBlender(False, 5, prec)
    mb.set_deferred_level0("copy")
    seen = []
    for on in (False, True, False):
        mb.set_narrow_copies(on)
        tiles = _tiles(rng, sizes, "bytes")
        od, om = _oracle_blend(oracle, 5, prec, corners, sizes, tiles)
        d, m = _gpu_blend(gpu, mb, prec, corners, sizes, tiles)
        assert np.array_equal(m, om) and np.array_equal(d, od), on
        seen.append(mb.feed_path()["narrowed"])
    assert seen == ["none", "confirmed", "none"], seen
    import torch
    mb.set_narrow_copies(True)
    tiles = _tiles(rng, sizes, "bytes")
    mb.prepare(corners, sizes)
    mb.feed(torch.from_numpy(tiles[0][0]).cuda(), torch.from_numpy(tiles[0][1]).cuda(), corners[0])
    with pytest.raises(Exception):
        mb.set_narrow_copies(False)
    mb.feed(torch.from_numpy(tiles[1][0]).cuda(), torch.from_numpy(tiles[1][1]).cuda(), corners[1])
    od, om = _oracle_blend(oracle, 5, prec, corners, sizes, tiles)
    d, m = mb.blend(out_f32=prec != I16)
    assert np.array_equal(_np(m), om) and np.array_equal(_np(d), od)
