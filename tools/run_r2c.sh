cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_blend.py tests/test_gpu_configs.py tests/test_gpu_config5.py tests/test_gpu_end_to_end.py tests/test_gpu_fuzz_slice.py tests/test_gpu_feather.py -x -q 2>&1 | tail -3)
python tools/pipeline_probe.py 0 5 2>&1 | grep "ms/pair\|collapse_gather_final\|pyr_down"
python bench.py --precision i16 --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
