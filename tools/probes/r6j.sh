O=gpurun_out/r6j; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt; cat $O/pytest.txt
ISX_FUZZ_ONLY=case_round6_calls python tools/fuzz_parity.py 240 611 $O/round6_fuzz_round6_calls_240s_seed611.json 2>&1 | tail -8
python tools/fuzz_parity.py 1200 20261006 $O/round6_fuzz_1200s_seed20261006.json 2>&1 | tail -4
python tools/determinism_soak.py 3000 $O/round6_determinism_3000.json 2>&1 | tail -3
python bench.py 2>/dev/null | grep "^{" | tail -1 > $O/bench_n1_final.json; python -c "
import json; d=json.load(open('$O/bench_n1_final.json')); print('n1', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_serialised'))"
