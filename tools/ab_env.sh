#!/usr/bin/env bash
# Same-box A/B of environment-selected variants of ONE library build (run on the GPU box via gpurun):
#   VARS="ISX_ROLL=0|ISX_ROLL=1,ISX_ROLL_R=7|ISX_ROLL=1" bash tools/ab_env.sh [bench args]
# three alternations; prints Mpix/s, ms per step and the serialised per-kernel times of one step.
IFS='|' read -ra VS <<< "${VARS:-ISX_ROLL=0|ISX_ROLL=1}"
for r in 1 2 3; do
  for v in "${VS[@]}"; do
    e=$(env $(echo "$v" | tr ',' ' ') python bench.py --no-cpu-baseline --no-dropin --no-live-traffic "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_one_step']
print(d['value'], d['ms_per_step'], 'final', (k.get('collapse_roll') or k.get('collapse_gather_final') or {}).get('ms'), 'roof', d['roofline']['avg_launch_ms'], d['roofline']['frac'], 'mid', k.get('collapse_gather',{}).get('ms'), 'pd0', (k.get('pyr_down0') or k.get('pyr_down_l0') or {}).get('ms'), 'pd', k.get('pyr_down',{}).get('ms'), 'warp', (k.get('warp_tile') or k.get('warp_img_mask') or {}).get('ms'))")
    echo "[$v] $e"
  done
done
