#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC counters (run on the GPU box).

    python tools/measure_traffic.py profiles/round2_traffic.json [-- bench args]

Runs bench.py twice under `rocprofv3 --kernel-trace --pmc X` — FETCH_SIZE and WRITE_SIZE in SEPARATE
passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2; MI355X_MICROARCH.md "rocprofv3 PMC slots") —
and writes, per kernel, the mean over its dispatches of
    traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-byte fabric requests as 64 bytes
for wide coalesced streaming reads, hence the doubling the guide prescribes (MI355X_MICROARCH.md "HBM").
WRITE_SIZE and narrow access widths are uncalibrated per the guide: treat the figure as +-2x on the
read side, exact enough to tell re-read waste from algorithmic traffic.
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else None


# kernel function name -> the launch name bench.py / the library's profiler reports
ALIAS = {"k_collapse_gather": None, "k_warp_img_mask": "warp_img_mask", "k_warp_tile": "warp_tile", "k_roi_scan": "roi_scan",
         "k_lap_acc_all": "lap_acc_all", "k_pyr_down": None, "k_pyr_down_multi": None, "k_collapse": None}


def launch_name(kname, full):
    if kname in ("k_collapse_roll", "k_collapse_roll_batch"):       # the last collapse step without LDS (launch names = kernel names without the k_)
        return "collapse_roll"
    if kname == "k_collapse_gather":
        return "collapse_gather_final" if re.search(r"k_collapse_gather<\d+, -?\d+, true", full) else "collapse_gather"
    if kname == "k_pyr_down0":           # level 0 -> 1 of CV_8UC3 / CV_16SC3 tiles
        return "pyr_down0"
    if kname in ("k_pyr_down", "k_pyr_down_multi"):
        return "pyr_down" if re.search(r"<\d+, -1>", full) else "pyr_down_l0"
    if kname == "k_collapse":
        return "collapse_final" if re.search(r"k_collapse<\d+, true", full) else "collapse"
    return ALIAS.get(kname) or kname


def run_pass(counter, extra):
    d = tempfile.mkdtemp(prefix="isx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-dropin", "--no-live-traffic"] + extra
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=240)
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] == counter:
                acc[launch_name(k, r["Kernel_Name"])].append(float(r["Counter_Value"]))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    out = sys.argv[1]
    extra = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
    fetch = run_pass("FETCH_SIZE", extra)
    write = run_pass("WRITE_SIZE", extra)
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        res[k] = {"FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "traffic_bytes": int((2 * f + w) * 1024)}
    json.dump({"note": "per launch; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction)",
               "bench_args": extra, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
