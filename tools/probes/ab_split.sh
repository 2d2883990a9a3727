# A/B of the level-1 step split in two launches (ISX_SPLIT1): alternating runs of the default bench line, then other precisions / configs / graph
B="python bench.py --steps 200 --warmup 5 --no-dropin --no-cpu-baseline --no-live-traffic"
P='import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], (d.get("two_steps_in_flight") or {}).get("Mpix_s"))'
run() { echo -n "[$1] $2 : " >> gpurun_out/ab_split.txt; env $1 $B $2 2>/dev/null | python -c "$P" >> gpurun_out/ab_split.txt; }
for rep in 1 2 3 4; do for v in ISX_SPLIT1=0 ISX_SPLIT1=1; do run $v ""; done; done
for rep in 1 2; do for v in ISX_SPLIT1=0 ISX_SPLIT1=1; do
  run $v "--precision i16"; run $v "--tile-type s16"; run $v "--graph"; run $v "--pairs 4 --streams 4"
  run $v "--kind spherical --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --tiles 8 --steps 30"
done; done
cat gpurun_out/ab_split.txt
