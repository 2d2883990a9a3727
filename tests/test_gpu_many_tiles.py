"""More than 20 tiles in ONE deferred multi-band cycle (BASELINE config 4's shape: a long row of tiles in one panorama).

A launch's kernel arguments hold 20 tiles' descriptors; until round 4 the 21st feed() ended the deferred cycle and the whole panorama went
through the eager destination pyramid.  Round 4: blend() cut the result into column strips that at most 20 tiles reach and ran the deferred
chain per strip (run_blend_deferred_strips; still there as ISX_TAB=0).  Round 5: ONE chain over all tiles, their descriptors in a table in device
memory (TileTab, cycle "deferred_table") - also for tiles stacked more than 20 deep over one place, which used to fall back to the eager cycle.
Here: the whole mosaic equals the oracle's, in the three precisions, for CV_8UC3 and CV_16SC3 tiles, with references and with private copies, on
both paths; a column window; stacks; the table's uploads (none in the steady state of a fixed rig)."""
import os

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


@pytest.fixture
def tab_env():
    """ISX_TAB is read per blend(): '0' selects round 4's column strips"""
    old = os.environ.get("ISX_TAB")
    yield lambda v: os.environ.__setitem__("ISX_TAB", v)
    if old is None:
        os.environ.pop("ISX_TAB", None)
    else:
        os.environ["ISX_TAB"] = old


@pytest.mark.parametrize("prec", [I16, F32, F16])
@pytest.mark.parametrize("s16", [False, True])
@pytest.mark.parametrize("tab", ["1", "0"])
def test_row_of_many_tiles_runs_as_one_chain_or_in_column_strips(gpu, oracle, tab_env, prec, s16, tab):
    import torch
    tab_env(tab)
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
        assert path["cycle"] == ("deferred_table" if tab == "1" else "deferred_strips"), path
        d, m = d.cpu().numpy(), m.cpu().numpy()
        assert np.array_equal(m, om), (mode, np.argwhere(m != om)[:4])
        assert np.array_equal(d, od), (mode, np.argwhere(d != od)[:4])


@pytest.mark.parametrize("tab", ["1", "0"])
def test_many_tiles_in_a_column_window(gpu, oracle, tab_env, tab):
    """A caller's own window (one rank's strip of a long panorama) of a cycle of more than 20 tiles: the table's chain on the window's columns
    (ISX_TAB=0: cut further into strips by the library)."""
    import torch
    tab_env(tab)
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


def _stack(rng, n, step):
    corners = [(int(rng.integers(0, 30)), step * i) for i in range(n)]        # n rows over the same columns
    sizes = [(int(rng.integers(80, 110)), 60) for _ in range(n)]
    tiles = [(rng.integers(0, 256, (h, w, 3)).astype(np.uint8), (rng.random((h, w)) > 0.2).astype(np.uint8) * 255) for (w, h) in sizes]
    return corners, sizes, tiles


@pytest.mark.parametrize("prec", [I16, F32])
@pytest.mark.parametrize("tab", ["1", "0"])
def test_tiles_stacked_beyond_twenty_over_one_place(gpu, oracle, tab_env, prec, tab):
    """24 rows of tiles over the same columns: every 128-column strip is reached by more than 20 tiles.  The table's chain takes them (up to ten
    tiles deep in one pixel: k_collapse_gather runs the last step); round 4's strips could not and fell back to the eager cycle."""
    import torch
    tab_env(tab)
    rng = np.random.default_rng(9)
    corners, sizes, tiles = _stack(rng, 24, 25)
    ob = oracle.MultiBand(3, prec)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)       # the caller's convertTo(CV_16S), W:294
    od, om = ob.blend(prec != I16)
    mb = gpu.MultiBandBlender(False, 3, prec)
    mb.set_deferred_level0(True)
    mb.prepare(corners, sizes)
    keep = []
    for (img, mask), c in zip(tiles, corners):
        ti, tm = torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()
        keep.append((ti, tm))
        mb.feed_u8(ti, tm, c)
    d, m = mb.blend(out_f32=prec != I16)
    assert mb.last_path()["cycle"] == ("deferred_table" if tab == "1" else "eager")
    assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)


def test_a_stack_too_deep_for_the_int16_shortcut_leaves_the_table(gpu, oracle):
    """The int16 last step of CV_8UC3 tiles does not issue the short accumulator's wrap: 128 tiles of +-255 cannot reach it.  140 tiles over one
    place can; such a cycle is kept off the table (strips, then the eager cycle) and still equals the oracle."""
    import torch
    rng = np.random.default_rng(11)
    n = 140
    corners = [(int(rng.integers(0, 6)), int(rng.integers(0, 6))) for _ in range(n)]
    sizes = [(40, 30)] * n
    tiles = [(rng.integers(0, 256, (30, 40, 3)).astype(np.uint8), np.full((30, 40), 255, np.uint8)) for _ in range(n)]
    ob = oracle.MultiBand(2, I16)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)
    od, om = ob.blend(False)
    mb = gpu.MultiBandBlender(False, 2, I16)
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
    # the same stack in fp32 has no such guard and stays on the table
    ob = oracle.MultiBand(2, F32)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)
    od, om = ob.blend(True)
    mb = gpu.MultiBandBlender(False, 2, F32)
    mb.set_deferred_level0(True)
    mb.prepare(corners, sizes)
    for (ti, tm), c in zip(keep, corners):
        mb.feed_u8(ti, tm, c)
    d, m = mb.blend(out_f32=True)
    assert mb.last_path()["cycle"] == "deferred_table"
    assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)


def test_the_table_is_uploaded_once_for_a_fixed_rig(gpu, oracle):
    """The tile tables travel in kernel arguments, chunk by chunk, and only the chunks that differ from what the device holds: the second and
    third blend() of the same rig on the same buffers upload nothing, a moved tile uploads again; results equal the oracle's every time."""
    import torch
    rng = np.random.default_rng(21)
    corners, sizes, tiles = _row_of_tiles(rng, 30, False)
    dev = [(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda()) for img, mask in tiles]
    mb = gpu.MultiBandBlender(False, 5, F32)
    mb.set_deferred_level0(True)
    w0 = None
    ups = []
    for rep in range(4):
        cs = list(corners)
        if rep == 3:
            cs[7] = (cs[7][0] + 4, cs[7][1] - 3)
        ob = oracle.MultiBand(5, F32)
        ob.prepare(cs, sizes)
        mb.prepare(cs, sizes)
        for (img, mask), (ti, tm), c in zip(tiles, dev, cs):
            ob.feed(img.astype(np.int16), mask, c)
            mb.feed_u8(ti, tm, c)
        d, m = mb.blend(out_f32=True)
        od, om = ob.blend(True)
        assert mb.last_path()["cycle"] == "deferred_table"
        assert np.array_equal(m.cpu().numpy(), om) and np.array_equal(d.cpu().numpy(), od)
        ups.append(mb.table_uploads())
    assert ups[0] > 0 and ups[1] == ups[0] and ups[2] == ups[0] and ups[3] > ups[2], ups


def test_a_table_cycle_captured_as_a_hipgraph_survives_another_blend(gpu, oracle):
    """A planned step of 24 tiles captured into a hipGraph: the graph holds the writes of its tile tables (a capture uploads every piece and vouches
    for none), so a replay equals the eager step even after another cycle of the same blender - other corners - rewrote the tables in between."""
    import torch
    from imagestitch_amd import synth
    from imagestitch_amd.pipeline import MosaicStitcher
    K, Rs = synth.camera_ring(640, 360, 1800.0, 24, 0.17)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    imgs = [torch.randint(0, 256, (360, 640, 3), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(24)]
    p = MosaicStitcher(imgs, K, Rs, 1800.0, "cylindrical", 4, F32, 0, None, "int16")
    ref = [t.clone() for t in p.step()]
    assert p.blender.last_path()["cycle"] == "deferred_table"
    p.capture()
    out, m = p.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref[0]) and torch.equal(m, ref[1])
    # another cycle of the same blender, the tiles moved: its tables overwrite the captured step's in device memory
    corners = [(c[0] + 2 * i, c[1]) for i, c in enumerate(p.corners)]
    with torch.cuda.stream(p.gstream):
        p.blender.prepare(corners, p.sizes)
        for i in range(24):
            p.blender.feed_u8(p.warped[i], p.seam[i], corners[i])
        d2, m2 = p.blender.blend(out_f32=True)
    torch.cuda.synchronize()
    ob = oracle.MultiBand(4, F32)
    ob.prepare(corners, p.sizes)
    for i in range(24):
        ob.feed(p.warped[i].cpu().numpy().astype(np.int16), p.seam[i].cpu().numpy(), corners[i])
    od, om = ob.blend(True)
    assert np.array_equal(d2.cpu().numpy(), od) and np.array_equal(m2.cpu().numpy(), om)
    out, m = p.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref[0]) and torch.equal(m, ref[1])
    # ... and the other order: the replay has just rewritten the device's tables behind the host's back - an eager blend of the moved
    # tiles right after it must not trust its mirror of what the device held before the replay (it compared equal and skipped the upload)
    with torch.cuda.stream(p.gstream):
        p.blender.prepare(corners, p.sizes)
        for i in range(24):
            p.blender.feed_u8(p.warped[i], p.seam[i], corners[i])
        d3, m3 = p.blender.blend(out_f32=True)
    torch.cuda.synchronize()
    assert np.array_equal(d3.cpu().numpy(), od) and np.array_equal(m3.cpu().numpy(), om)
    out, m = p.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref[0]) and torch.equal(m, ref[1])
    p.check_plan()
