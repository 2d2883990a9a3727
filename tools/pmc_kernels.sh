#!/usr/bin/env bash
# Per-kernel VALU utilisation of one 4K pair (run on the GPU box via gpurun): tools/pmc_kernels.sh [precision] [tag]
#   VALU busy = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x kernel duration x 2.4 GHz); occupancy = SQ_WAVE_CYCLES x 4 / (1024 x duration x 2.4 GHz)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PREC=${1:-1}; TAG=${2:-pmc}
P=gpurun_out/$TAG; rm -rf $P; mkdir -p $P
timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $P/run -- python tools/pipeline_probe.py $PREC 5 > $P/log.txt 2>&1
python - "$P" <<'PY'
import csv, glob, collections, re, sys
P=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(P+'/run/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n: continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for f in glob.glob(P+'/run/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n: continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print('%-44s %7s %8s %9s %9s %7s %7s %6s' % ('kernel','us','waves','valu/wave','vmem/wave','valu%','occ','wait%'))
for k,cs in sorted(acc.items(), key=lambda kv:-sum(dur[kv[0]])):
    a={c:sum(v)/len(v) for c,v in cs.items()}
    d=sorted(dur[k])[len(dur[k])//2]
    w=max(a.get('SQ_WAVES',1),1)
    cyc=d*1e-6*2.4e9*1024
    print('%-44s %7.1f %8d %9.0f %9.1f %6.0f%% %7.2f %5.0f%%' % (k[:44], d, w, a['SQ_INSTS_VALU']/w, a['SQ_INSTS_VMEM']/w, 100*a['SQ_INSTS_VALU']*4/cyc, a['SQ_WAVE_CYCLES']*4/cyc, 100*a['SQ_WAIT_INST_ANY']/max(a['SQ_WAVE_CYCLES'],1)))
PY
