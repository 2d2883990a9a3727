#!/usr/bin/env bash
# Where k_collapse_top2's time goes: timing-only builds of blend.hip that return after each phase (-DT2_ABL=n; wrong pixels).
#   build here:  bash tools/probes/top2_ablation.sh build       on the GPU box:  gpurun -- 'bash tools/probes/top2_ablation.sh run'
cd "$(dirname "$0")/../.."
if [ "${1:-run}" = build ]; then
  for n in 8 1 2 3; do bash tools/build_variant.sh t2abl$n blend.hip "-DT2_ABL=$n" & done; wait
  for n in 4 5 0; do bash tools/build_variant.sh t2abl$n blend.hip "-DT2_ABL=$n" & done; wait
else
  L=imagestitch_amd/csrc/libimagestitch_hip.so; cp $L /tmp/lib_keep.so
  for r in 1 2; do for n in 8 1 2 3 4 5 0; do cp tmp_ab/libt2abl$n.so $L
    python bench.py --no-cpu-baseline --no-dropin --no-live-traffic --steps 40 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_one_step']
print('T2_ABL=$n', 'collapse_top', k['collapse_top']['ms'], 'step', d['ms_per_step'])"; done; done
  cp /tmp/lib_keep.so $L
fi
