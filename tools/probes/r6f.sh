O=gpurun_out/r6f; mkdir -p $O
ISX_WARP_XG=4 python -m pytest tests/test_gpu_warp.py tests/test_gpu_blend.py tests/test_gpu_configs.py tests/test_gpu_strips.py -x -q 2>&1 | tail -3
python tools/probes/warp_under_blend_probe.py 200 2>&1 | grep -v amdgpu.ids | tee $O/warp_under_blend.txt
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --no-live-traffic "$@" 2>/dev/null | grep "^{" | tail -1 > $O/bench_$name.json; python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
b c3_graph_2br --pairs 16 --batch --graph --streams 2
b c3_graph_4br --pairs 16 --batch --graph --streams 4
b c3_graph_4br_bs2 --pairs 16 --batch --graph --streams 4 --batch-size 2
b c3_graph_8br --pairs 16 --batch --graph --streams 8
b c3_graph_16br --pairs 16 --batch --graph --streams 16
b c3_batch_4streams --pairs 16 --batch --streams 4
b c3_batch_2streams --pairs 16 --batch --streams 2
