cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
P=gpurun_out/r2d; rm -rf $P; mkdir -p $P
for v in v2r1 v2r2 v1; do
  case $v in v2r1) export ISX_WARP_ROWS=1; unset ISX_WARP_V1;; v2r2) export ISX_WARP_ROWS=2; unset ISX_WARP_V1;; v1) export ISX_WARP_V1=1;; esac
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $P/$v -- python tools/warp_probe.py 20 > $P/$v.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC --output-format csv -d $P/${v}b -- python tools/warp_probe.py 20 > $P/${v}b.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
for v in ['v2r1','v2r1b','v2r2','v2r2b','v1','v1b']:
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/r2d/%s/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if 'k_warp' not in n: continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(v, {c: round(sum(x)/len(x)) for c,x in acc.items()})
PY
