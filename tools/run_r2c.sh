cd $GRAFT_REPO_ROOT
for dbg in 0 16 4; do ISX_WARP_DBG=$dbg python tools/warp_probe.py 2000 2>&1 | tail -1; done
(timeout 900 python -m pytest tests/test_gpu_warp.py -x -q 2>&1 | tail -2)
