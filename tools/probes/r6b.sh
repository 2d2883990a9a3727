O=gpurun_out/r6b; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest.txt
python bench.py 2>$O/bench.err | grep "^{" | tail -1 > $O/bench_n1.json
python bench.py --no-cpu-baseline --no-dropin --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/bench_many_tiles_64.json
python bench.py --no-cpu-baseline --no-dropin --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2 2>/dev/null | grep "^{" | tail -1 > $O/bench_many_tiles_24.json
python tools/probes/roi_latency_probe.py > $O/roi_latency.txt 2>&1; ISX_ROI_HOST=0 python tools/probes/roi_latency_probe.py >> $O/roi_latency.txt 2>&1
python tools/probes/literal_host_probe.py 1 > $O/literal_host_time.txt 2>&1
ISX_ROI_HOST=0 python tools/probes/literal_host_probe.py 1 > $O/literal_host_time_devroi.txt 2>&1
cat $O/pytest.txt; cat $O/roi_latency.txt $O/literal_host_time.txt
