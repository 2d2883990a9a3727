O=gpurun_out/r6c; mkdir -p $O
python -m pytest tests/test_gpu_many_tiles.py -x -q 2>&1 | tail -40 > $O/pytest_many.txt
python bench.py --no-cpu-baseline --no-dropin --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/bench_many_tiles_64.json
python bench.py --no-cpu-baseline --no-dropin --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/bench_many_tiles_24.json
ISX_ROI_HOST=0 python bench.py --no-cpu-baseline --no-dropin --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/bench_many_tiles_64_devroi.json
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_many_tiles.py 2>&1 | tail -4 > $O/pytest.txt
cat $O/pytest_many.txt $O/pytest.txt
python -c "
import json
for n in ('64','24','64_devroi'):
    m=json.load(open('$O/bench_many_tiles_%s.json'%n)); print(n, m['value'], m['ms_per_step'])
"
