#!/bin/bash
# Kernel trace of a batched step (bench.py --pairs P --batch): prints, for one step of the timed region, every launch in start order with
# its duration, the gap to the previous launch's end on the same queue, and the busy / idle time of the step (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/trace_batch; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --no-cpu-baseline --no-dropin --no-live-traffic --steps 6 --warmup 3 "$@" > $OUT/bench.json 2> $OUT/err.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "at::" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the last-collapse launches: take the launches between the 3rd-last and 2nd-last "final" launch groups
fin = [i for i, r in enumerate(rows) if "k_collapse_roll_batch" in r["Kernel_Name"] or "k_collapse_gather_batch<1, 0, true" in r["Kernel_Name"]]
if not fin:
    fin = [i for i, r in enumerate(rows) if "k_collapse_roll" in r["Kernel_Name"] or ("k_collapse_gather" in r["Kernel_Name"] and ", true," in r["Kernel_Name"])]
# group consecutive finals of one step: a step ends at a final followed by a warp
ends = [i for k, i in enumerate(fin) if k + 1 == len(fin) or any("k_warp" in rows[j]["Kernel_Name"] for j in range(i + 1, fin[k + 1]))]      # a step's last final launch
a, b = ends[-3] + 1, ends[-2] + 1
t0 = int(rows[a]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows[a:b])
busy = collections.Counter()
last_end = {}
iv = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    short = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print("%9.1f us  +%7.1f  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end[q]) / 1e3 if q in last_end else 0.0, q, short))
    last_end[q] = e; busy[short.split("<")[0]] += e - s; iv.append((s, e))
iv.sort(); cover = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e: cover += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cover += cur_e - cur_s
print("step: %.1f us wall, %.1f us with at least one kernel running (idle %.1f us); per-kernel busy sums (us): %s" % ((t1 - t0) / 1e3, cover / 1e3, (t1 - t0 - cover) / 1e3, {k: round(v / 1e3, 1) for k, v in busy.items()}))
PY
