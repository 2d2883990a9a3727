#!/bin/bash
# rocprofv3 averages of the step's kernels for library variants tmp_ab/lib<V>.so (same box, alternating):  VARS="head new" bash tools/probes/ab_kernel_avg.sh [bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
L=imagestitch_amd/csrc/libimagestitch_hip.so; cp $L /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do for v in $VARS; do
  cp $R/tmp_ab/lib$v.so $R/$L; rm -rf /tmp/kavg; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kavg -- python $R/bench.py --no-cpu-baseline --no-dropin --no-live-traffic --steps 200 --warmup 10 "$@" > /tmp/kavg.json 2>/dev/null
  python - "$v" <<'PY'
import csv, glob, sys, json
f = glob.glob("/tmp/kavg/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def avg(sub):
    r = [x for x in rows if sub in x["Name"]]
    return sum(float(x["TotalDurationNs"]) for x in r) / max(sum(int(x["Calls"]) for x in r), 1) / 1e3
print("[%s]" % sys.argv[1], " ".join("%s %.2f" % (n, avg(s)) for n, s in (("warp", "k_warp_tile"), ("pd0", "k_pyr_down0"), ("pd", "k_pyr_down_multi"), ("top", "k_collapse_top"), ("mid", "k_collapse_gather"), ("roll", "k_collapse_roll"))))
PY
done; done
cp /tmp/lib_keep.so $R/$L
