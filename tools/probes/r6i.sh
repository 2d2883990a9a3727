O=gpurun_out/r6i; mkdir -p $O
python -m pytest tests/test_gpu_warp.py -x -q 2>&1 | tail -5
for r in 1 2 3; do for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu-baseline --no-dropin --no-live-traffic 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', d['value'], d['ms_per_step'])"; done; done
python bench.py --no-cpu-baseline --no-dropin --no-live-traffic 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default env', d['value'], d['ms_per_step'])"
env | grep -i -E "^HIP_|^HSA_|^GPU_|^AMD_|^ROC" | head -20
