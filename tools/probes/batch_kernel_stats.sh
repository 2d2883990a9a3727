#!/bin/bash
# Per-kernel averages (rocprofv3 --kernel-trace --stats) of the batched 16-pair step against the pair-by-pair step.   bash tools/probes/batch_kernel_stats.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/batch_stats; mkdir -p $O
for v in "batch:--pairs 16 --batch" "batch12:--pairs 16 --batch --batch-size 16" "single:--pairs 1"; do
  n=${v%%:*}; a=${v#*:}
  rm -rf /tmp/prof_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o t -- python $R/bench.py $a --steps 20 --no-dropin --no-cpu-baseline --no-live-traffic > $O/$n.json 2> $O/$n.err
  f=$(find /tmp/prof_$n -name '*kernel_stats.csv' | head -1)
  python - "$f" "$n" <<'PY' > $O/$n.stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[2])
for r in rows[:16]:
    print('%-90s %6s %10.1f us  %5s %%' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
  cat $O/$n.stats.txt; python -c "import json,sys; d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
