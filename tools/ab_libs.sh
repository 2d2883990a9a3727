#!/usr/bin/env bash
# Same-box A/B of library builds (run on the GPU box via gpurun): tmp_ab/lib<V>.so for every V in $VARS is copied over the in-tree
# library in turn, $REPS alternations (default 2); prints Mpix/s, ms per step, the last collapse step's HIP-event time in the timed region.
L=imagestitch_amd/csrc/libimagestitch_hip.so
cp $L /tmp/lib_keep.so
for r in $(seq 1 ${REPS:-2}); do
  for v in $VARS; do
    cp tmp_ab/lib$v.so $L
    e=$(python bench.py --no-cpu-baseline --no-dropin --no-live-traffic "$@" 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_one_step']
print(d['value'], d['ms_per_step'], 'final(serialised)', (k.get('collapse_roll') or k.get('collapse_gather_final') or {}).get('ms'), 'dominant', d['roofline']['kernel'], d['roofline']['avg_launch_ms'], 'warp', (k.get('warp_tile') or k.get('warp_img_mask') or {}).get('ms'), 'pd0', (k.get('pyr_down0') or k.get('pyr_down_l0') or {}).get('ms'), 'pd', (k.get('pyr_down') or {}).get('ms'), 'mid', (k.get('collapse_gather') or {}).get('ms'), 'top', (k.get('collapse_top') or {}).get('ms'))")
    echo "[$v] $e"
  done
done
cp /tmp/lib_keep.so $L
