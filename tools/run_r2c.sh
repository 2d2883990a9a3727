cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_blend.py tests/test_gpu_config5.py tests/test_gpu_configs.py -x -q 2>&1 | tail -15)
for dbg in 0 4 7; do ISX_WARP_DBG=$dbg python tools/warp_probe.py 2000 2>&1 | tail -1; done
ISX_WARP_V1=1 python tools/warp_probe.py 2000 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_one_step'];print(d['value'], d['ms_per_step'], 'warp',k['warp_img_mask']['ms'])"
