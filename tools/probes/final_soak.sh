# the round's closing soak on the GPU box: a long fuzz run on the final kernels and the single pair eager / as a hipGraph in alternation
O=gpurun_out/final; mkdir -p $O
for r in 1 2 3; do for v in "" "--graph"; do python bench.py --no-cpu-baseline --no-dropin --no-live-traffic $v 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[single pair $v]', d['value'], d['ms_per_step'])"; done; done > $O/round6_graph_vs_eager_ab.txt 2>&1
cat $O/round6_graph_vs_eager_ab.txt
python tools/fuzz_parity.py ${1:-2400} 77001 $O/round6_fuzz_${1:-2400}s_seed77001.json 2>&1 | tail -3
