#!/bin/bash
# Kernel trace of the default bench command; prints the launches of ONE planned step of the timed region in order, with start offsets,
# durations and the gaps between them (run on the GPU box; output under gpurun_out/trace/).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --no-cpu-baseline --no-dropin "$@" > $OUT/bench.json 2> $OUT/err.log
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last occurrence of the final collapse kernel that is preceded by a complete step
idx = [i for i, n in enumerate(names) if "k_collapse_roll<" in n or ("k_collapse_gather<" in n and ", true, false>" in n)]
end = idx[-3]
start = end
while start > 0 and "k_warp_tile" not in names[start]:
    start -= 1
while start > 0 and ("k_warp_tile" in names[start - 1] or "roi" in names[start - 1]):
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev_end = None
for r in rows[start:end + 4]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    short = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%9.1f us  +%7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0, r.get("Queue_Id", "?"), short))
    prev_end = e
PY
