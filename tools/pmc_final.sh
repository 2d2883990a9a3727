#!/usr/bin/env bash
# PMC counters of the pair's kernels (run on the GPU box via gpurun): tools/pmc_final.sh "<counters>"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=gpurun_out/pmc_final; rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $P/run -- python tools/pipeline_probe.py 1 5 sync > $P/log.txt 2>&1
python - <<'PY'
import csv, glob, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_final/run/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n: continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(acc.items()):
    print('%-40s'%k, {c: round(sum(v)/len(v)) for c,v in cs.items()})
PY
