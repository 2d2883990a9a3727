#!/usr/bin/env bash
# tests + probe + per-kernel VALU counts (run on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/pipeline_probe.py 1 5 2>&1 | tail -10
python tools/pipeline_probe.py 0 5 2>&1 | grep ms/pair
if [ "$1" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; P=gpurun_out/prof3; rm -rf $P; mkdir -p $P
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM --output-format csv -d $P/sq -- python tools/pipeline_probe.py 1 5 > $P/sq.log 2>&1
  python - <<'PY'
import csv, glob, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/prof3/sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n: continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in sorted(acc.items()):
    a={c:sum(v)/len(v) for c,v in cs.items()}
    w=a.get('SQ_WAVES',1)
    print('%-38s waves %7d valu/wave %6.0f salu/wave %5.0f vmem/wave %5.1f valu_busy_us %6.1f wave_life_us %5.1f' % (k,w,a['SQ_INSTS_VALU']/w,a['SQ_INSTS_SALU']/w,a['SQ_INSTS_VMEM']/w,a['SQ_ACTIVE_INST_VALU']*4/1024/2100,a['SQ_WAVE_CYCLES']*4/w/2100))
PY
fi
