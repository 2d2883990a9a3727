#!/usr/bin/env bash
# Same-box A/B of library builds (run on the GPU box via gpurun): tmp_ab/lib<V>.so for every V in $VARS is copied over the
# in-tree library in turn, three alternations, eager and hipGraph bench lines each.  Usage:
#   VARS="V0 V1" bash tools/ab_bench.sh        (tmp_ab/ is scratch, not tracked)
L=imagestitch_amd/csrc/libimagestitch_hip.so
VARS=${VARS:-"V0 V1"}
for r in 1 2 3; do
  for v in $VARS; do
    cp tmp_ab/lib$v.so $L
    e=$(python bench.py --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());k=d['kernels_ms_one_step'];print(d['value'], 'final',(k.get('collapse_roll') or k['collapse_gather_final'])['ms'], 'mid',k['collapse_gather']['ms'],'pd0',(k.get('pyr_down0') or k['pyr_down_l0'])['ms'],'pd',k['pyr_down']['ms'],'warp',(k.get('warp_tile') or k['warp_img_mask'])['ms'])")
    g=$(python bench.py --graph --no-cpu-baseline --no-dropin 2>/dev/null | grep "^{" | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'])")
    echo "$v eager $e graph $g"
  done
done
cp tmp_ab/lib$(echo $VARS | cut -d' ' -f1).so $L
