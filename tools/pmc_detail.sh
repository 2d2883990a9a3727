#!/usr/bin/env bash
# Where the cycles of the pyramid kernels go: three SQ counter passes over one 4K pair (run on the GPU box): tools/pmc_detail.sh [tag]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-pmc_detail}
P=gpurun_out/$TAG; rm -rf $P; mkdir -p $P
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $P/run$n -- python tools/pipeline_probe.py 1 5 > $P/log$n.txt 2>&1; }
pass 1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pass 2 SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU
pass 3 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH
pass 4 SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
python - "$P" <<'PY'
import csv, glob, collections, re, sys
P=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(P+'/run*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if 'at::' in n or 'rocclr' in n: continue
        m=re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    a={c:sum(v)/len(v) for c,v in cs.items()}
    w=max(a.get('SQ_WAVES',1),1)
    if w < 2000: continue
    wc=max(a.get('SQ_WAVE_CYCLES',1),1)
    print(k)
    print('   waves %d; per wave: cycles %.0f (x4 = clocks), wait_any %.0f%%, wait_inst_any %.0f%%, wait_inst_lds %.0f%%' % (w, wc/w, 100*a.get('SQ_WAIT_ANY',0)/wc, 100*a.get('SQ_WAIT_INST_ANY',0)/wc, 100*a.get('SQ_WAIT_INST_LDS',0)/wc))
    print('   insts per wave: valu %.0f salu %.0f smem %.0f lds %.0f vmem_rd %.0f vmem_wr %.0f branch %.0f' % tuple(a.get(c,0)/w for c in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_SMEM','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR','SQ_INSTS_BRANCH')))
    b=max(a.get('SQ_BUSY_CYCLES',1),1)
    print('   active-inst cycles / busy cycles: any %.2f valu %.2f sca %.2f lds %.2f vmem %.2f flat %.2f misc %.2f; inst_cycles_vmem/wave %.0f salu/wave %.0f' % tuple([a.get(c,0)/b for c in ('SQ_ACTIVE_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_SCA','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_VMEM','SQ_ACTIVE_INST_FLAT','SQ_ACTIVE_INST_MISC')]+[a.get('SQ_INST_CYCLES_VMEM',0)/w, a.get('SQ_INST_CYCLES_SALU',0)/w]))
    print('   lds: bank_conflict/idx_active %.2f, data_fifo_full %.0f cmd_fifo_full %.0f per wave; ta addr_fifo_full %.0f cmd_fifo_full %.0f wr_data_fifo_full %.0f per wave' % (a.get('SQ_LDS_BANK_CONFLICT',0)/max(a.get('SQ_LDS_IDX_ACTIVE',1),1), a.get('SQ_LDS_DATA_FIFO_FULL',0)/w, a.get('SQ_LDS_CMD_FIFO_FULL',0)/w, a.get('SQ_VMEM_TA_ADDR_FIFO_FULL',0)/w, a.get('SQ_VMEM_TA_CMD_FIFO_FULL',0)/w, a.get('SQ_VMEM_WR_TA_DATA_FIFO_FULL',0)/w))
PY
