cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_warp.py tests/test_gpu_configs.py tests/test_gpu_fuzz_slice.py -x -q 2>&1 | tail -3)
python tools/warp_probe.py 2000 2>&1 | tail -1
