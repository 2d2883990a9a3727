cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/bench_r2c.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r2c.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print({k:v['ms'] for k,v in d['kernels_ms_one_step'].items()})
for k,v in d['dropin'].items(): print(k, v)
PY
