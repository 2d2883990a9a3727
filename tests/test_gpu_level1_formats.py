"""Level 1 of the deferred cycle has two layouts (round 4): 16-byte records, and - when k_collapse_roll runs the last step - dense 12-byte
image records for out_1 plus PLANAR tile levels (12-byte image records + a weight plane).  The layout is chosen per process
(ISX_OUT12 / ISX_G1P), so the comparison runs tests/helpers/level1_formats_digest.py in two child processes: every blend of its set
(single and batched chains, three precisions, both tile types, 2 / 5 / 7 bands, a column window) must hash the same in both.  (That
each of them equals the oracle is what the rest of the suite shows with the defaults.)  Round 5 added a third form of the planar level 1 (Q8
records, ISX_G1Q8), compared the same way."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, kinds=("single", "batch", "window")):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "level1_formats_digest.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.split(" ")[0] in kinds]


def test_level1_layouts_give_the_same_bits(gpu):
    new = _run({"ISX_OUT12": "1", "ISX_G1P": "1"})
    old = _run({"ISX_OUT12": "0", "ISX_G1P": "0"})
    only_out = _run({"ISX_OUT12": "1", "ISX_G1P": "0"})
    only_g1 = _run({"ISX_OUT12": "0", "ISX_G1P": "1"})
    assert len(new) >= 66 and len(new) == len(old) == len(only_out) == len(only_g1)
    # the last step of most of the set is k_collapse_roll (path 3): the set exercises the new layouts
    assert sum(1 for ln in new if ln.startswith("single") and ln.split(" ")[6] == "collapse_roll") >= 36
    for a, b, c, d in zip(new, old, only_out, only_g1):
        assert a == b == c == d, (a, b, c, d)


def test_q8_level1_gives_the_same_bits(gpu):
    """Round 5: level 1 of fp32 pyramids over CV_8UC3 tiles as three unsigned shorts (k / 256 exactly) - on by default, ISX_G1Q8=0 turns it off."""
    on = _run({}, ("single", "batch", "window", "fmt"))
    off = _run({"ISX_G1Q8": "0"}, ("single", "batch", "window", "fmt"))
    dig = lambda lines: [ln for ln in lines if not ln.startswith("fmt")]
    assert dig(on) == dig(off)
    fmt_on = [ln.split(" ") for ln in on if ln.startswith("fmt")]
    fmt_off = [ln.split(" ") for ln in off if ln.startswith("fmt")]
    # (fmt prec s16 bands rig format): Q8 only for fp32 (prec 1) over CV_8UC3 tiles, and only with the switch on
    assert sum(1 for f in fmt_on if f[5] == "planar_q8") >= 4
    assert all(f[1] == "1" and f[2] == "0" for f in fmt_on if f[5] == "planar_q8")
    assert all(f[5] == "planar_q8" for f in fmt_on if f[1] == "1" and f[2] == "0" and f[5] != "records")
    assert not any(f[5] == "planar_q8" for f in fmt_off)
    assert any(f[5] == "planar" for f in fmt_off if f[1] == "1" and f[2] == "0")


def _np(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else a


@pytest.mark.parametrize("layout", ["pair", "three_deep", "ragged"])
@pytest.mark.parametrize("bands", [2, 5])
def test_q8_level1_on_extreme_bytes_equals_the_oracle(gpu, oracle, layout, bands):
    """Q8 records hold 256 x a level-1 channel as an unsigned short: the largest value is 255 * 256 = 65 280 (a tile of 255s), runs of 0 and 255 next to
    each other give every k in between; odd and tiny tiles put rims everywhere.  The caller's tiles (references), private copies (level 1 written by
    feed()) and CV_16SC3 tiles that are bytes (narrowed copies) must all take Q8 records in fp32 and equal the oracle bit for bit; CV_16SC3 tiles that are
    NOT bytes after all (one value outside 0..255) take the widened cycle: level 1 produced again as floats, same bits as the oracle's."""
    import numpy as np
    import torch
    rng = np.random.default_rng({"pair": 51, "three_deep": 52, "ragged": 53}[layout] + bands)
    if layout == "pair":
        corners, sizes = [(-40, 7), (233, -12)], [(411, 300), (397, 290)]
    elif layout == "three_deep":
        corners, sizes = [(0, 0), (150, 9), (290, -6), (430, 4)], [(420, 260), (420, 255), (420, 262), (300, 250)]
    else:
        corners, sizes = [(3, 1), (57, -2), (9, 77)], [(71, 93), (5, 140), (131, 33)]
    tiles = []
    for i, (w, h) in enumerate(sizes):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.int64).astype(np.uint8)
        if i == 0:
            img[:] = 255                                   # k = 65 280 everywhere
            img[h // 3:, w // 2:] = 0
        else:
            img[::3, ::2] = 255
            img[1::5, 1::7] = 0
        mask = rng.integers(0, 256, (h, w), dtype=np.int64).astype(np.uint8)
        mask[rng.random((h, w)) < 0.5] = 255
        mask[rng.random((h, w)) < 0.1] = 0
        tiles.append((img, mask))
    ob = oracle.MultiBand(bands, 1)
    ob.prepare(corners, sizes)
    for (img, mask), c in zip(tiles, corners):
        ob.feed(img.astype(np.int16), mask, c)
    od, om = ob.blend(True)
    seen_q8 = 0
    for mode, kind in ((True, "u8"), ("copy", "u8"), ("copy", "s16")):
        mb = gpu.MultiBandBlender(False, bands, 1)
        mb.set_deferred_level0(mode)
        mb.prepare(corners, sizes)
        keep = []
        for (img, mask), c in zip(tiles, corners):
            ti = torch.from_numpy(img if kind == "u8" else img.astype(np.int16)).cuda()
            tm = torch.from_numpy(mask).cuda()
            keep.append((ti, tm))
            mb.feed(ti, tm, c)
            if mode == "copy":
                ti.fill_(9), tm.fill_(99)
        d, m = mb.blend(out_f32=True)
        lp, fmt = mb.last_path(), mb.level1_format()
        assert np.array_equal(_np(m), om), (layout, mode, kind)
        assert np.array_equal(_np(d), od), (layout, mode, kind, np.argwhere(_np(d) != od)[:4])
        # (the rolling kernel as the last step = planar level 1 = Q8 records here; tiles too small for it: the gathering kernel and 16-byte records)
        assert fmt == ("planar_q8" if lp["last_step"] == "collapse_roll" else "records"), (layout, mode, kind, lp, fmt)
        seen_q8 += fmt == "planar_q8"
    assert layout != "pair" or seen_q8 == 3, (layout, seen_q8)
    # not bytes after all: one short outside 0..255 in the second tile
    img2 = tiles[1][0].astype(np.int16)
    img2[min(7, img2.shape[0] - 1), min(3, img2.shape[1] - 1), 1] = 300
    ob = oracle.MultiBand(bands, 1)
    ob.prepare(corners, sizes)
    for i, ((img, mask), c) in enumerate(zip(tiles, corners)):
        ob.feed(img2 if i == 1 else img.astype(np.int16), mask, c)
    od, om = ob.blend(True)
    mb = gpu.MultiBandBlender(False, bands, 1)
    mb.set_deferred_level0("copy")
    mb.prepare(corners, sizes)
    keep = []
    for i, ((img, mask), c) in enumerate(zip(tiles, corners)):
        ti, tm = torch.from_numpy(img2 if i == 1 else img.astype(np.int16)).cuda(), torch.from_numpy(mask).cuda()
        keep.append((ti, tm))
        mb.feed(ti, tm, c)
    d, m = mb.blend(out_f32=True)
    assert np.array_equal(_np(m), om) and np.array_equal(_np(d), od), (layout, "violated", np.argwhere(_np(d) != od)[:4])
    assert mb.level1_format() != "planar_q8", mb.level1_format()
    assert mb.feed_path()["narrowed"] in ("widened", "none"), mb.feed_path()
