# one panorama of 24 / 64 4K tiles in a row on one GPU: column strips of the deferred chain (default) against the eager cycle (ISX_STRIPS=0, what > 20 tiles took before)
P='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], d["config"].get("path"))'
for args in "--tiles 24 --focal 9000 --yaw 0.12" "--tiles 64 --focal 24000 --yaw 0.046"; do
for v in ISX_STRIPS=1 ISX_STRIPS=0; do
  echo -n "[$v] $args : " >> gpurun_out/ab_many_tiles.txt
  env $v timeout 600 python bench.py $args --steps 6 --warmup 2 --no-dropin --no-cpu-baseline --no-live-traffic 2> gpurun_out/ab_many_tiles.err | python -c "$P" >> gpurun_out/ab_many_tiles.txt 2>&1
  tail -3 gpurun_out/ab_many_tiles.err >> gpurun_out/ab_many_tiles.txt
done; done
cat gpurun_out/ab_many_tiles.txt
