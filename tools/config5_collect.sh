cd $GRAFT_REPO_ROOT; O=gpurun_out/round4; TAG=round4; mkdir -p $O/strips
line() { grep "^{" | tail -1; }
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | line > $O/${TAG}_bench_$name.json; }
b config5_ring8_8k --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2
b config5_8k_pair --kind spherical --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 10 --warmup 3
ISX_ROLL_R23=0 python bench.py --no-cpu-baseline --no-dropin --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2 2>/dev/null | line > $O/${TAG}_bench_config5_ring8_8k_ISX_ROLL_R23_0.json
C5="--kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 10 --warmup 3"
for w in 2 4 8; do for r in $(seq 0 $((w-1))); do python bench.py $C5 --strip-of $r/$w 2>/dev/null | line > $O/strips/config5_strip_${r}_of_$w.json; done; done
python - "$O" "$TAG" <<'PY'
import json, sys
O, TAG = sys.argv[1], sys.argv[2]
whole = json.load(open("%s/%s_bench_config5_ring8_8k.json" % (O, TAG)))
out = {"what": "BASELINE config 5 as ONE panorama cut into N column strips: the time of every rank's share, each measured alone on one MI355X with "
               "`python bench.py <config 5 flags> --strip-of R/N` (no gather; a one-GPU box cannot run N ranks)",
       "whole_on_one_gpu": {"ms_per_step": whole["ms_per_step"], "Mpix_s": whole["value"]}, "ranks": {}}
for w in (2, 4, 8):
    rs = [json.load(open("%s/strips/config5_strip_%d_of_%d.json" % (O, r, w))) for r in range(w)]
    out["ranks"][str(w)] = {"ms_per_step": [r["ms_per_step"] for r in rs], "tiles": [r["config"]["tiles_this_rank"] for r in rs],
                            "window": [r["config"]["window"] for r in rs], "panorama_cols": rs[0]["config"]["panorama_cols"],
                            "compute_speedup_vs_one_gpu": round(whole["ms_per_step"] / max(r["ms_per_step"] for r in rs), 2)}
json.dump(out, open("%s/%s_strips_config5.json" % (O, TAG), "w"), indent=1)
PY
for f in config5_ring8_8k config5_8k_pair config5_ring8_8k_ISX_ROLL_R23_0; do python -c "
import json; d=json.load(open('$O/${TAG}_bench_$f.json')); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'])"; done
python -c "
import json; d=json.load(open('$O/${TAG}_strips_config5.json')); print({k:(max(v['ms_per_step']), v['compute_speedup_vs_one_gpu']) for k,v in d['ranks'].items()})"
