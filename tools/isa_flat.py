#!/usr/bin/env python3
"""Per-kernel count of FLAT and scratch memory instructions in a gfx950 assembly listing.

    tools/isa_flat.py file.s [name-substring ...]      (file.s from `hipcc -S --cuda-device-only`)
    tools/isa_flat.py build/blend.o [name-substring ...] (a built object: its gfx950 code object is unbundled and disassembled, ~3 s -
                                                         this is the shipped binary, what tests/test_isa_flat.py checks)

A FLAT access is what the compiler emits for a pointer whose address space it could not prove: it is counted on
lgkmcnt as well as vmcnt and drains LDS waits (csrc/collapse_roll.inc explains the trap).  Hot kernels must show 0.
Prints one line per kernel that has any (or every kernel matching the substrings), and a JSON summary last.
"""
import collections
import json
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


LLVM = "/opt/rocm/lib/llvm/bin/"


def disassemble(obj, workdir):
    """gfx950 code object of a host object built by hipcc (section .hip_fatbin, an offload bundle) -> path of its llvm-objdump -d listing"""
    import os
    fat, co, dis = (os.path.join(workdir, n) for n in ("fat.bin", "dev.co", "dev.dis"))
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co], check=True)
    with open(dis, "w") as f:
        subprocess.run([LLVM + "llvm-objdump", "-d", co], stdout=f, check=True)
    return dis


def scan(path):
    res = collections.OrderedDict()
    cur = None
    if path.endswith(".o"):
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            return scan_listing(disassemble(path, d), re.compile(r"^[0-9a-f]+ <(_Z\S+)>:"), True)
    return scan_listing(path, re.compile(r"^(_Z\S+):\s*; @"), False)


def scan_listing(path, rx_k, objdump):
    res = collections.OrderedDict()
    cur = None
    rx_i = re.compile(r"^\s+((?:flat|scratch|global|buffer|ds)_[a-z0-9_]+)")
    with open(path, errors="replace") as f:
        for line in f:
            m = rx_k.match(line)
            if m:
                cur = res.setdefault(m.group(1), collections.Counter())
                continue
            if cur is None:
                continue
            if "s_endpgm" in line and not objdump:      # (a listing of an object has padding s_endpgm / s_code_end inside; the next label ends a kernel)
                cur = None
                continue
            m = rx_i.match(line)
            if m:
                op = m.group(1)
                if op.startswith("flat_load"): cur["flat_load"] += 1
                elif op.startswith("flat_store"): cur["flat_store"] += 1
                elif op.startswith("flat_atomic"): cur["flat_atomic"] += 1
                elif op.startswith("scratch_"): cur["scratch"] += 1
                elif op.startswith("global_load"): cur["global_load"] += 1
                elif op.startswith("global_store"): cur["global_store"] += 1
    return res


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    res = scan(path)
    dm = demangle(list(res))
    summary = {}
    for k, c in res.items():
        name = dm[k]
        if pats and not any(p in name for p in pats):
            continue
        if not pats and not (c["flat_load"] or c["flat_store"] or c["scratch"]):
            continue
        short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
        short = re.sub(r"\((?!anonymous).*$", "", short)
        summary[short] = dict(c)
        print("%-70s flat_load %3d flat_store %3d scratch %3d global_load %3d global_store %3d" % (short[:70], c["flat_load"], c["flat_store"], c["scratch"], c["global_load"], c["global_store"]))
    print(json.dumps({"kernels": len(summary), "flat_load": sum(v.get("flat_load", 0) for v in summary.values()),
                      "flat_store": sum(v.get("flat_store", 0) for v in summary.values()), "scratch": sum(v.get("scratch", 0) for v in summary.values())}))


if __name__ == "__main__":
    main()
