cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
(timeout 900 python -m pytest tests/test_gpu_warp.py tests/test_gpu_blend.py tests/test_gpu_config5.py -x -q 2>&1 | tail -15) > gpurun_out/r2b/pytest.log 2>&1
cat gpurun_out/r2b/pytest.log
for wr in 1 2 4; do
  echo "WR=$wr"; ISX_WARP_ROWS=$wr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_one_step'];print(d['value'], d['ms_per_step'], 'warp',k['warp_img_mask']['ms'])"
done
echo V1; ISX_WARP_V1=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_one_step'];print(d['value'], d['ms_per_step'], 'warp',k['warp_img_mask']['ms'])"
