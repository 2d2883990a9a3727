# A/B of the 12-byte out_1 records (ISX_OUT12) and the planar level 1 (ISX_G1P): alternating runs of the default bench line, then the other precisions / configs
B="python bench.py --steps 200 --warmup 5 --no-dropin --no-cpu-baseline --no-live-traffic"
P='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"])'
run() { echo -n "[$1] $2 : " >> gpurun_out/ab_planar.txt; env $1 $B $2 2>/dev/null | python -c "$P" >> gpurun_out/ab_planar.txt; }
for rep in 1 2 3 4; do
for v in "ISX_G1P=0 ISX_OUT12=0" "ISX_G1P=0 ISX_OUT12=1" "ISX_G1P=1 ISX_OUT12=1"; do run "$v" ""; done; done
for rep in 1 2; do
for v in "ISX_G1P=0 ISX_OUT12=0" "ISX_G1P=0 ISX_OUT12=1" "ISX_G1P=1 ISX_OUT12=1"; do
  run "$v" "--precision i16"; run "$v" "--tile-type s16"; run "$v" "--precision f16acc32"
  run "$v" "--kind spherical --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --tiles 8 --steps 30"
done; done
cat gpurun_out/ab_planar.txt
