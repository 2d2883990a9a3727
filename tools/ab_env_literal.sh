#!/usr/bin/env bash
# Same-box A/B of environment-selected variants on the drop-in legs: VARS="A=1|B=2,C=3" [MODES="literal sync"] [PREC=1] bash tools/ab_env_literal.sh
IFS='|' read -ra VS <<< "${VARS}"
for r in $(seq 1 ${REPS:-2}); do
  for v in "${VS[@]}"; do
    for m in ${MODES:-literal sync}; do
      o=$(env $(echo "$v" | tr ',' ' ') python tools/pipeline_probe.py ${PREC:-1} 5 $m 2>/dev/null | python -c "
import sys
ms=''; k={}
for l in sys.stdin:
    p=l.split()
    if l.startswith('ms/pair'): ms=p[1]
    elif 'launches/step' in l: k[p[0]]=float(p[4])*1e3
print(ms, ' '.join('%s %.1f' % (n, k[n]) for n in ('feed_strip','feed_pd0','collapse_roll','pyr_down0','feed_copy') if n in k))")
      echo "[$v] $m $o"
    done
  done
done
