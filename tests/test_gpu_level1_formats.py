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
