"""Column strips of one panorama on CPU: the host logic of mosaic.py, the oracle-level proof of the rule that picks a
strip's tiles (with ONLY those tiles fed, the oracle's MultiBandBlender gives the strip's columns of the whole blend unchanged), and
the world-2 gloo assembly of strips into the panorama."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagestitch_amd import mosaic, _lib


def test_strip_windows():
    for width, world in ((5576, 2), (1606, 6), (1000, 8), (128, 4), (40000, 8), (129, 2)):
        wins, sw = mosaic.strip_windows(width, world, _lib.WINDOW_GRANULE)
        assert len(wins) == world and sw % _lib.WINDOW_GRANULE == 0 and sw * world >= width
        assert wins[0][0] == 0 and all(x0 % _lib.WINDOW_GRANULE == 0 for x0, _ in wins)
        covered = 0
        for x0, x1 in wins:
            assert x1 - x0 in (0, sw) and (x1 > x0) == (x0 < width)
            if x1 > x0:
                assert x0 == covered
                covered = x1
        assert covered >= width and covered - width < sw
    assert mosaic.strip_windows(1606, 6)[0][-1] == (1920, 1920)      # a strip that starts past the right edge is empty


def test_feed_rect_follows_multibandblender_feed():
    # dst_roi 0..1024 x 0..256 (L = 5: multiples of 32), tile at (300, 10) 200 x 100: gap 96 -> [204, 596) snapped to 192, width 404 -> 416
    assert mosaic.feed_rect((0, 0, 1024, 256), 5, (300, 10), (200, 100)) == (192, 0, 416, 224)
    # a tile at the right edge: the padded rectangle is shifted back inside dst_roi
    x, y, w, h = mosaic.feed_rect((0, 0, 1024, 256), 5, (900, 0), (124, 256))
    assert x + w == 1024 and w % 32 == 0 and x % 32 == 0 and x <= 900 - 96 and (y, h) == (0, 256)
    # relative to dst_roi's corner
    assert mosaic.feed_rect((-500, -20, 1024, 256), 5, (-200, -10), (200, 100)) == mosaic.feed_rect((0, 0, 1024, 256), 5, (300, 10), (200, 100))


def test_window_needs_recursion():
    lo, hi = mosaic.window_needs(3, 256, 384, [1024, 512, 256, 128])
    assert (lo, hi) == ([256, 127, 62, 30], [384, 193, 98, 50])
    lo, hi = mosaic.window_needs(2, 0, 4096, [1024, 512, 256])      # clipped to the levels
    assert (lo, hi) == ([0, 0, 0], [1024, 512, 256])


def test_tile_columns_for_window():
    corners = [(-37 + i * 130, 5 + (i % 3) * 4) for i in range(7)]
    sizes = [(200, 96)] * 7
    fw = corners[-1][0] + 200 - corners[0][0]
    # the whole result needs every column of every tile
    assert mosaic.tile_columns_for_window(corners, sizes, 3, 0, fw) == {i: (0, 200) for i in range(7)}
    cols = mosaic.tile_columns_for_window(corners, sizes, 3, 384, 512)
    assert sorted(cols) == mosaic.tiles_for_window(corners, sizes, 3, 384, 512)
    for i, (c0, c1) in cols.items():
        assert 0 <= c0 < c1 <= 200
        # a tile that lies wholly inside the window's reach is needed whole; one that straddles its edge is not
        x_lo, x_hi = corners[i][0] - corners[0][0], corners[i][0] - corners[0][0] + 200
        if x_lo >= 384 and x_hi <= 512:
            assert (c0, c1) == (0, 200)
    assert any(c1 - c0 < 200 for c0, c1 in cols.values())
    assert mosaic._reflect([-3, -1, 0, 4, 5, 7, 12], 5).tolist() == [2, 0, 0, 4, 4, 2, 2]      # fedcba|abcde|edcba (OpenCV BORDER_REFLECT)


def _row_of_tiles(n, tw, th, step, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    corners = [(-37 + i * step, 5 + (i % 3) * 4) for i in range(n)]
    imgs = [rng.integers(0, 256, (th, tw, 3), dtype=np.uint8) for _ in range(n)]
    return corners, imgs


@pytest.mark.parametrize("bands,prec_name", [(3, "I16"), (4, "F32"), (5, "I16")])
def test_only_the_listed_tiles_reach_a_strip_oracle(bands, prec_name):
    """The rule blend()'s window rests on, checked against the oracle alone: feed the oracle's MultiBandBlender ONLY the tiles
    tiles_for_window lists for a strip and the strip's columns come out exactly as in the blend of all tiles - and for at least
    one strip a tile outside the list does change other columns (the lists are not trivially everything)."""
    from imagestitch_amd import synth
    from oracle import capi as O
    n, tw, th, step = 7, 200, 96, 130
    corners, imgs = _row_of_tiles(n, tw, th, step, 77)
    sizes = [(tw, th)] * n
    masks = synth.seam_masks(corners, [np.full((th, tw), 255, np.uint8)] * n)
    prec = getattr(O, prec_name)

    def blend(active):
        mb = O.MultiBand(bands, prec)
        mb.prepare(corners, sizes)
        for i in active:
            mb.feed(imgs[i].astype(np.int16), masks[i], corners[i])
        return mb.blend(prec != O.I16)

    full, full_mask = blend(range(n))
    fw = full.shape[1]
    wins, _ = mosaic.strip_windows(fw, 4)
    subsets = 0
    for x0, x1 in wins:
        if x1 == x0:
            continue
        act = mosaic.tiles_for_window(corners, sizes, bands, x0, x1)
        subsets += len(act) < n
        part, part_mask = blend(act)
        xe = min(x1, fw)
        assert np.array_equal(part[:, x0:xe], full[:, x0:xe]), (x0, x1, act)
        assert np.array_equal(part_mask[:, x0:xe], full_mask[:, x0:xe])
        if len(act) < n:
            assert not np.array_equal(part, full)
    assert subsets >= 2


def test_assemble_strips_layout():
    rows, sw, width, world = 5, 128, 300, 3
    pano = torch.arange(rows * width * 3, dtype=torch.int32).reshape(rows, width, 3)
    padded = torch.zeros((rows, world * sw, 3), dtype=torch.int32)
    padded[:, :width] = pano
    strips = torch.stack([padded[:, r * sw:(r + 1) * sw].reshape(-1) for r in range(world)])
    assert torch.equal(mosaic.assemble_strips(strips, rows, sw, width), pano)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    from imagestitch_amd import synth
    from oracle import capi as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, tw, th, step, bands = 5, 160, 64, 110, 3
    corners, imgs = _row_of_tiles(n, tw, th, step, 5)
    sizes = [(tw, th)] * n
    masks = synth.seam_masks(corners, [np.full((th, tw), 255, np.uint8)] * n)

    def blend(active):
        mb = O.MultiBand(bands, O.I16)
        mb.prepare(corners, sizes)
        for i in active:
            mb.feed(imgs[i].astype(np.int16), masks[i], corners[i])
        return mb.blend(False)[0]

    fw, fh = (max(c[0] for c in corners) + tw) - min(c[0] for c in corners), (max(c[1] for c in corners) + th) - min(c[1] for c in corners)
    wins, sw = mosaic.strip_windows(fw, world)
    x0, x1 = wins[rank]
    # this rank's strip, from its own tiles only (the CPU stand-in for the windowed HIP blend: same columns by the test above)
    mine = blend(mosaic.tiles_for_window(corners, sizes, bands, x0, x1))
    send = torch.zeros((fh, sw, 3), dtype=torch.int16)
    send[:, :min(x1, fw) - x0] = torch.from_numpy(mine[:, x0:min(x1, fw)].copy())
    got = mosaic.gather_mosaics(send.reshape(-1))
    pano = mosaic.assemble_strips(got, fh, sw, fw).numpy()
    ret[rank] = bool(np.array_equal(pano, blend(range(n))))
    dist.barrier()
    dist.destroy_process_group()


def test_strips_assemble_the_panorama_world2_gloo():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_one_interval_per_tile_bounds_the_reach_rule_from_outside():
    """The library's own strips (run_blend_deferred_strips in csrc/blend.hip: a cycle of more than 20 tiles is blended in column strips) plan with
    ONE interval per tile instead of the level-by-level rule: a window [x0, x1) takes a tile iff x0 < fx + fw + 3 * 2^L and x1 > fx - 5 * 2^L.  That
    must be a SUPERSET of tiles_for_window (extra tiles change no pixel of the window, a missing one would): checked on random rigs and windows."""
    rng = np.random.default_rng(12)
    checked = 0
    for _ in range(300):
        n = int(rng.integers(2, 40))
        bands = int(rng.integers(1, 8))
        sizes = [(int(rng.integers(4, 900)), int(rng.integers(4, 300))) for _ in range(n)]
        corners = [(int(rng.integers(-3000, 3000)), int(rng.integers(-100, 100))) for _ in range(n)]
        c, s = np.asarray(corners), np.asarray(sizes)
        tl, br = c.min(0), (c + s).max(0)
        w, h = int(br[0] - tl[0]), int(br[1] - tl[1])
        L = min(bands, int(np.ceil(np.log(float(max(w, h))) / np.log(2.0))))
        m = 1 << L
        roi = (int(tl[0]), int(tl[1]), w + (m - w % m) % m, h + (m - h % m) % m)
        for _ in range(6):
            x0 = int(rng.integers(0, max(w // 128, 1))) * 128
            x1 = min(x0 + int(rng.integers(1, 12)) * 128, w)
            if x1 <= x0:
                continue
            exact = set(mosaic.tiles_for_window(corners, sizes, bands, x0, x1))
            bound = set()
            for i in range(n):
                fx, _, fw, _ = mosaic.feed_rect(roi, L, corners[i], sizes[i])
                if x0 < fx + fw + 3 * m and x1 > fx - 5 * m:
                    bound.add(i)
            assert exact <= bound, (corners, sizes, bands, x0, x1, exact - bound)
            checked += 1
    assert checked > 1000
