O=gpurun_out/r6h; mkdir -p $O
python -m pytest tests/test_gpu_warp.py tests/test_gpu_many_tiles.py tests/test_gpu_configs.py tests/test_gpu_strips.py tests/test_gpu_dist.py -x -q 2>&1 | tail -5 > $O/pytest.txt; cat $O/pytest.txt
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin --no-live-traffic "$@" 2>/dev/null | grep "^{" | tail -1 > $O/bench_$name.json; python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'])"; }
b many_tiles_64 --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2
b many_tiles_24 --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2
ISX_WARP_BATCH=0 python bench.py --no-cpu-baseline --no-dropin --no-live-traffic --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('many_tiles_64 ISX_WARP_BATCH=0', d['value'], d['ms_per_step'])"
b c3_batch_graph --pairs 16 --batch --graph
b c3_batch_3streams --pairs 16 --batch --streams 3
b c3_batch_4streams --pairs 16 --batch --streams 4
b c3_streams4 --pairs 16
b c5_ring8 --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2
b i16 --precision i16
b f16 --precision f16acc32
b n1
