"""The shipped gfx950 code has no FLAT access in a kernel that matters.

A FLAT load is what the compiler emits for a pointer whose address space it cannot prove - here: pointers LOADED from the device-resident tile
table (TileTab, more than 20 tiles in one mosaic) instead of arriving in the kernel arguments.  It is counted on lgkmcnt as well as vmcnt, so every
LDS wait drains the global loads too (csrc/collapse_roll.inc tells the story for pointers that went through an asm statement).  Round 5's *_tab
kernels issued 10 - 54 of them each; `as_global` (csrc/blend.hip) types the table's pointers where they are loaded.  This test disassembles the
code object inside the BUILT objects (imagestitch_amd/csrc/build/*.o - what libimagestitch_hip.so was linked from; ~3 s per object, no GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "imagestitch_amd", "csrc", "build")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _scan(obj):
    import isa_flat
    path = os.path.join(BUILD, obj)
    if not os.path.exists(path):
        subprocess.check_call(["bash", os.path.join(ROOT, "imagestitch_amd", "csrc", "build.sh")])
    res = isa_flat.scan(path)
    names = isa_flat.demangle(list(res))
    return {names[k]: v for k, v in res.items()}


def test_no_flat_access_in_any_blend_kernel():
    res = _scan("blend.o")
    tab = [n for n in res if "_tab<" in n]
    assert len(tab) >= 40, len(tab)          # the table forms of pyrDown (level 0 and above), the top, the level steps and the last step, per precision
    hot = [n for n in res if any(k in n for k in ("k_collapse_roll<", "k_pyr_down0<", "k_collapse_top2<", "k_feed_strip<", "k_collapse_gather<", "k_pyr_down_multi<"))]
    assert len(hot) >= 20, len(hot)
    bad = {n: dict(c) for n, c in res.items() if c["flat_load"] or c["flat_store"] or c["flat_atomic"]}
    assert not bad, bad


def test_the_table_form_of_a_kernel_issues_the_loads_of_its_argument_form():
    """same body, same loads: the table view only changes where the descriptors come from (s_load from the table instead of the argument segment)"""
    res = _scan("blend.o")
    import re
    pairs = 0
    for n, c in res.items():
        m = re.match(r"(.*k_collapse_roll)_tab<(.*)>\(", n)
        if not m:
            continue
        twin = next((v for k, v in res.items() if k.startswith(m.group(1) + "<" + m.group(2) + ">(")), None)
        if twin is None:
            continue
        pairs += 1
        assert c["global_load"] == twin["global_load"] and c["global_store"] == twin["global_store"], (n, dict(c), dict(twin))
    assert pairs >= 3, pairs


def test_warp_kernels_keep_flat_accesses_out_of_line():
    res = _scan("warp.o")
    bad = {n: dict(c) for n, c in res.items() if (c["flat_load"] or c["flat_store"]) and not any(k in n for k in ("slow_bilinear", "slow_nearest"))}
    assert not bad, bad        # (the generic per-pixel samplers are out-of-line functions that take generic pointers by design: DESIGN.md §3)
