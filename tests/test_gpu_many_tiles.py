"""More than 20 tiles in ONE deferred multi-band cycle (BASELINE config 4's shape: a long row of tiles in one panorama).

A launch's kernel arguments hold 20 tiles' descriptors; until round 4 the 21st feed() ended the deferred cycle and the whole panorama went
through the eager destination pyramid.  Now blend() cuts the result into column strips that at most 20 tiles reach and runs the deferred
chain per strip (run_blend_deferred_strips: the rule of isx_blender_set_window applied by the library itself).  Every strip equals the same
columns of the whole blend bit for bit - here: the whole mosaic equals the oracle's, in the three precisions, for CV_8UC3 and CV_16SC3 tiles,
with references and with private copies; and tiles stacked in so many rows that one strip is reached by more than 20 fall back to the eager
cycle with the same result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

I16, F32, F16 = 0, 1, 2


def _row_of_tiles(rng, n, s16):
    corners, sizes, tiles = [], [], []
    x = 0
    for i in range(n):
        w, h = int(rng.integers(70, 120)), int(rng.integers(60, 90))
        corners.append((x, int(rng.integers(-8, 9))))
        sizes.append((w, h))
        x += int(w * rng.uniform(0.35, 0.8))          # neighbours overlap, second neighbours sometimes
        if s16:
            img = rng.integers(-3000, 3001, (h, w, 3)).astype(np.int16)
        else:
            img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        mask = (rng.random((h, w)) > 0.15).astype(np.uint8) * 255
        mask[rng.random((h, w)) < 0.1] = 130
        tiles.append((img, mask))
    return corners, sizes, tiles


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("s16", [False, True])
def test_row_of_many_tiles_runs_in_column_strips(gpu, oracle, prec, s16):
    import torch
    rng = np.random.default_rng(100 + prec + 10 * int(s16))
    n = 47
    corners, sizes, tiles = _row_of_tiles(rng, n, s16)
    bands = 4
    ob = oracle.MultiBand(bands, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)       # the caller's convertTo(CV_16S), W:294
    f32 = prec != I16
    od, om = ob.blend(f32)
    for mode in (True, "copy"):
        mb = gpu.MultiBandBlender(False, bands, prec)
        mb.set_deferred_level0(mode)
        mb.prepare(corners, sizes)
        keep = []
        for (img, mask), c in zip(tiles, corners):
            ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
            keep.append((ti, tm))
            if s16:
                mb.feed(ti, tm, c)
            else:
                mb.feed_u8(ti, tm, c)
            if mode == "copy":
                ti.fill_(3), tm.fill_(9)
        d, m = mb.blend(out_f32=f32)
        path = mb.last_path()
        assert path["cycle"] == "deferred_strips", path
        d, m = d.cpu().numpy(), m.cpu().numpy()
        assert np.array_equal(m, om), (mode, np.argwhere(m != om)[:4])
        assert np.array_equal(d, od), (mode, np.argwhere(d != od)[:4])


def test_many_tiles_in_a_column_window(gpu, oracle):
    """A caller's own window (one rank's strip of a long panorama) of a cycle of more than 20 tiles: cut further by the library."""
    import torch
    rng = np.random.default_rng(7)
    corners, sizes, tiles = _row_of_tiles(rng, 40, False)
    ob = oracle.MultiBand(3, F32)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)       # the caller's convertTo(CV_16S), W:294
    od, om = ob.blend(True)
    mb = gpu.MultiBandBlender(False, 3, F32)
    mb.set_deferred_level0(True)
    mb.prepare(corners, sizes)
    w, h = mb.result_size()
    x0, x1 = 256, min(256 + 1536, (w // 128) * 128)
    mb.set_window(x0, x1)
    keep = []
    for (img, mask), c in zip(tiles, corners):
        ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
        keep.append((ti, tm))
        mb.feed_u8(ti, tm, c)
    d, m = mb.blend(out_f32=True)
    d, m = d.cpu().numpy(), m.cpu().numpy()
    assert d.shape[1] == x1 - x0
    assert np.array_equal(m, om[:, x0:x1]) and np.array_equal(d, od[:, x0:x1])


def test_tiles_stacked_beyond_the_strip_limit_fall_back_to_the_eager_cycle(gpu, oracle):
    import torch
    rng = np.random.default_rng(9)
    n = 24
    corners = [(int(rng.integers(0, 30)), 25 * i) for i in range(n)]        # 24 rows over the same columns
    sizes = [(int(rng.integers(80, 110)), 60) for _ in range(n)]
    tiles = [(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), (rng.random((h, w)) > 0.2).astype(np.uint8) * 255) for (w, h) in sizes]
    ob = oracle.MultiBand(3, I16)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)       # the caller's convertTo(CV_16S), W:294
    od, om = ob.blend(False)
    mb = gpu.MultiBandBlender(False, 3, I16)
    mb.set_deferred_level0(True)
    mb.prepare(corners, sizes)
    keep = []
    for (img, mask), c in zip(tiles, corners):
        ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
        keep.append((ti, tm))
        mb.feed_u8(ti, tm, c)
    d, m = mb.blend(out_f32=False)
    assert mb.last_path()["cycle"] == "eager"
    assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)
