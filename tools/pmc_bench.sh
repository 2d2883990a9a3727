#!/usr/bin/env bash
# SQ / TCC counters of the default bench step's kernels (run on the GPU box via gpurun), one rocprofv3 pass per counter group
# (never combined with a trace domain other than --kernel-trace):
#   tools/pmc_bench.sh <out.txt> "<counters pass 1>" ["<counters pass 2>" ...]      env: BENCH_ARGS, KFILTER (regex on the kernel name)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
: > $out
n=0
for grp in "$@"; do
  n=$((n+1)); P=/tmp/pmc_bench_$n; rm -rf $P
  timeout ${PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $P -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-dropin --no-live-traffic $BENCH_ARGS > /tmp/pmc_bench_$n.log 2>&1
  python - "$P" "${KFILTER:-k_}" >> $out <<'PY'
import csv, glob, collections, re, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n or not re.search(sys.argv[2], n): continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(acc.items()):
    print('%-44s'%k, {c: round(sum(v)/len(v)) for c,v in cs.items()}, 'n=%d'%len(next(iter(cs.values()))))
PY
done
cat $out
