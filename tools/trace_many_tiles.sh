#!/bin/bash
# Kernel trace of one many-tile panorama step (bench.py --tiles N ...): per queue the busy time, and on the union of all queues the gaps
# (intervals in which no kernel runs) of ONE step of the timed region - what stands between the step and its kernel sum.
#   gpurun -- 'bash tools/trace_many_tiles.sh <tag> --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python $R/bench.py --no-cpu-baseline --no-dropin "$@" > $OUT/bench.json 2> $OUT/err.log
python - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n): return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
names = [short(r["Kernel_Name"]) for r in rows]
# a step begins at the first k_warp_tile that follows a last-step kernel (ROI kernels of the side stream ignored)
starts, prev = [], None
for i, n in enumerate(names):
    if n.startswith("k_roi"): continue
    if n.startswith("k_warp_tile") and prev is not None and (prev.startswith("k_collapse_roll") or prev.startswith("k_collapse_gather")): starts.append(i)
    prev = n
if len(starts) < 4:
    print("too few steps found", len(starts)); sys.exit(0)
print("step to step: %.1f us" % ((int(rows[starts[-2]]["Start_Timestamp"]) - int(rows[starts[-3]]["Start_Timestamp"])) / 1e3))
a, b = starts[-3], starts[-2]
step = rows[a:b]
t0 = min(int(r["Start_Timestamp"]) for r in step); t1 = max(int(r["End_Timestamp"]) for r in step)
print("step of %d launches, %.1f us from first start to last end" % (len(step), (t1 - t0) / 1e3))
perq = collections.defaultdict(float); perk = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    perq[r.get("Queue_Id", "?")] += d; k = short(r["Kernel_Name"]); perk[k][0] += 1; perk[k][1] += d
print("busy per queue (us):", dict((q, round(v, 1)) for q, v in perq.items()))
for k, (c, d) in sorted(perk.items(), key=lambda kv: -kv[1][1]): print("  %-50s x%4d  %9.1f us" % (k, c, d))
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
busy = 0; cur_s, cur_e = iv[0]; gaps = []
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((cur_e, s)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("union busy %.1f us, idle %.1f us in %d gaps" % (busy / 1e3, (t1 - t0 - busy) / 1e3, len(gaps)))
hist = collections.Counter()
for s, e in gaps:
    g = (e - s) / 1e3
    hist["<2" if g < 2 else "2-5" if g < 5 else "5-10" if g < 10 else "10-30" if g < 30 else ">30"] += g
print("idle by gap size (us):", dict((k, round(v, 1)) for k, v in hist.items()))
# the 25 largest gaps with the kernels either side
big = sorted(gaps, key=lambda g: g[0] - g[1])[:25]
byend = {}
for i, r in enumerate(step): byend.setdefault(int(r["End_Timestamp"]), i)
bystart = {}
for i, r in enumerate(step): bystart.setdefault(int(r["Start_Timestamp"]), i)
for s, e in sorted(big):
    i, j = byend.get(s), bystart.get(e)
    print("  gap %7.1f us at +%8.1f : after %s (q%s) before %s (q%s)" % ((e - s) / 1e3, (s - t0) / 1e3, short(step[i]["Kernel_Name"]) if i is not None else "?", step[i].get("Queue_Id") if i is not None else "?",
                                                              short(step[j]["Kernel_Name"]) if j is not None else "?", step[j].get("Queue_Id") if j is not None else "?"))
PY
