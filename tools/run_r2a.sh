cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r2a/pytest.log 2>&1
(timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | tail -3) > gpurun_out/r2a/bench_default.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2a/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>&1 | tail -3) > gpurun_out/r2a/bench_traced.log 2>&1
(timeout 900 python bench.py --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2 --no-cpu-baseline --no-dropin 2>&1 | tail -3) > gpurun_out/r2a/bench_config5_ring8.log 2>&1
cat gpurun_out/r2a/pytest.log; tail -c 3000 gpurun_out/r2a/bench_default.log; tail -c 1500 gpurun_out/r2a/bench_config5_ring8.log
