cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_warp.py tests/test_gpu_configs.py tests/test_gpu_seam.py -x -q 2>&1 | tail -3)
VARS="V0 V1" bash tools/ab_bench.sh
