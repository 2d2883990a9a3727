#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this part for the access shapes of the pyramid kernels (run on the GPU box):
#   bash tools/fetch_calib.sh > gpurun_out/fetch_calib.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $R/tools/probes/fetch_calib.hip -o /tmp/fetch_calib || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fc_$C
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/fc_$C -- /tmp/fetch_calib > /tmp/fc_$C.log 2>&1
done
python3 - <<'PY'
import csv, glob
N = 1 << 30
vals = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/fc_%s/**/*counter_collection.csv" % C, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == C:
                vals.setdefault(r["Kernel_Name"].split("(")[0], {})[C] = float(r["Counter_Value"])
print("# tools/fetch_calib.sh on MI355X: every kernel touches each byte of a 1 GiB buffer once (Infinity Cache swept before each);")
print("# counter KiB * 1024 / 2^30 = what the counter reports per byte actually moved")
print("%-24s %14s %14s" % ("kernel", "FETCH_SIZE/byte", "WRITE_SIZE/byte"))
for k, v in vals.items():
    print("%-24s %14.3f %14.3f" % (k, v.get("FETCH_SIZE", 0) * 1024 / N, v.get("WRITE_SIZE", 0) * 1024 / N))
PY
