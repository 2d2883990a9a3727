#!/bin/bash
# The drop-in legs (device mats) of the default bench line under two environments, alternating on one box:  VARS="ISX_G1Q8=0|ISX_G1Q8=1" bash tools/probes/dropin_ab_env.sh
IFS='|' read -ra VS <<< "${VARS:-ISX_G1Q8=0|ISX_G1Q8=1}"
for r in 1 2; do
  for v in "${VS[@]}"; do
    env $(echo "$v" | tr ',' ' ') python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['dropin']
print('[$v]', d['value'], d['ms_per_step'], 'fused', p['fused_device_f32']['ms_per_pair'], p['fused_device_i16']['ms_per_pair'], 'literal', p['literal_device_f32']['ms_per_pair'], p['literal_device_i16']['ms_per_pair'], p['literal_device_f32'].get('feed_path'))"
  done
done
