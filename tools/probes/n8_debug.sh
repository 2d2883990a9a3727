#!/bin/bash
# the world-8 rehearsal of tests/test_gpu_dist.py::test_bench_walks_its_n8_path_on_one_gpu, N times, with the owner's diagnosis of a mismatched chunk
N=${1:-6}; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ISX_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 ISX_CHECK_DEBUG=1
for i in $(seq 1 $N); do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29500 + i)) bench.py --gpus 8 --steps 2 --warmup 2 --preflight-ms 0 \
    --gather-backend ${BACKEND:-p2p} --gather ${GATHER:-chunk} --check-gather --no-cpu-baseline --no-dropin --no-live-traffic --pairs 4 --width 640 --height 360 --focal 500 "$@" 2> /tmp/n8_err_$i.txt | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('run', $i, d['multi_gpu']['gather_check']['mismatched_over_all_ranks'], d['multi_gpu']['gather_check'].get('mismatched_by_rank'))"
  grep -h "px differ" /tmp/n8_err_$i.txt | head -4
done
