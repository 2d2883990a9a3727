#!/usr/bin/env bash
# timeline of the banded host warp: copies and kernels of the last iterations (rocprofv3 --kernel-trace --memory-copy-trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/duplex_trace; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python tools/probes/duplex_warp_probe.py > $O/log.txt 2>&1
python - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
ev = []
for f in glob.glob(O + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")), r.get("Size") or ""))   # columns vary by version
for f in glob.glob(O + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_warp" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel", ""))
ev.sort()
last = ev[-60:]
t0 = last[0][0]
for s, e, n, sz in last:
    print("%9.1f us  +%7.1f us  %-28s %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, sz))
PY
