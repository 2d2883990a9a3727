#!/usr/bin/env bash
# Same-box A/B of library builds on the drop-in legs (run on the GPU box via gpurun): tmp_ab/lib<V>.so for every V in $VARS copied over the in-tree
# library in turn, $REPS alternations; per run the literal / fused(sync) step of tools/pipeline_probe.py: ms per pair and the kernels that differ.
L=imagestitch_amd/csrc/libimagestitch_hip.so
cp $L /tmp/lib_keep.so
for r in $(seq 1 ${REPS:-2}); do
  for v in $VARS; do
    cp tmp_ab/lib$v.so $L
    for m in ${MODES:-literal sync}; do
      o=$(python tools/pipeline_probe.py ${PREC:-1} 5 $m 2>/dev/null | python -c "
import sys
ms=''; k={}
for l in sys.stdin:
    p=l.split()
    if l.startswith('ms/pair'): ms=p[1]
    elif 'launches/step' in l: k[p[0]]=float(p[4])*1e3
print(ms, ' '.join('%s %.1f' % (n, k[n]) for n in ('feed_strip','feed_pd0','collapse_roll','pyr_down','collapse_gather') if n in k))")
      echo "[$v] $m $o"
    done
  done
done
cp /tmp/lib_keep.so $L
