O=gpurun_out/r6e; mkdir -p $O
python -m pytest tests/test_gpu_configs.py -x -q 2>&1 | tail -30 > $O/pytest_cfg.txt
cat $O/pytest_cfg.txt
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --no-live-traffic "$@" 2>/dev/null | grep "^{" | tail -1 > $O/bench_$name.json; python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
b c3_batch_graph --pairs 16 --batch --graph
b c3_batch_graph_3br --pairs 16 --batch --graph --streams 3
b c3_batch_graph_4br --pairs 16 --batch --graph --streams 4
b c3_batch_3streams --pairs 16 --batch --streams 3
b c3_batch_one --pairs 16 --batch
b c3_streams4 --pairs 16
done
