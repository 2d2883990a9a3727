cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_blend.py::test_linear_pair_blend_4k tests/test_gpu_fuzz_slice.py -x -q 2>&1 | tail -15)
mkdir -p gpurun_out/fuzz
timeout 1000 python tools/fuzz_parity.py 600 7 gpurun_out/fuzz/round2_fuzz_600s_seed7.json 2>&1 | tail -5
