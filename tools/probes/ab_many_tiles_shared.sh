# one panorama of 24 / 64 4K tiles in a row on one GPU: the strips with the Gaussian chains produced once for all tiles (round 5, default) against per strip (ISX_STRIPS_SHARED=0, round 4)
P='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; k=d["kernels_ms_one_step"]
print(d["value"], d["ms_per_step"], "host_enqueue_ms_per_pair", d["config"]["host_enqueue_ms_per_pair"], d["config"].get("path"), {n: v["ms"] for n, v in k.items()})'
for rep in 1 2; do
for args in "--tiles 24 --focal 9000 --yaw 0.12" "--tiles 64 --focal 24000 --yaw 0.046"; do
for v in ISX_STRIPS_SHARED=1 ISX_STRIPS_SHARED=0; do
  echo -n "[$v] $args : " >> gpurun_out/ab_many_tiles_shared.txt
  env $v timeout 600 python bench.py $args --steps 6 --warmup 2 --no-dropin --no-cpu-baseline --no-live-traffic 2> gpurun_out/ab_many_tiles.err | python -c "$P" >> gpurun_out/ab_many_tiles_shared.txt 2>&1
done; done; done
cat gpurun_out/ab_many_tiles_shared.txt
