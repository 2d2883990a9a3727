#!/usr/bin/env bash
# Collects the measurement evidence of a round on the GPU box (run via gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles.sh round2 [fuzz_seconds]'
# Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
TAG=${1:-round6}; FUZZ=${2:-600}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
line() { grep "^{" | tail -1; }
# 1. HBM traffic per kernel (two PMC passes of their own), then the default line that reads it
python tools/measure_traffic.py $O/${TAG}_traffic.json > $O/traffic.log 2>&1
cp $O/${TAG}_traffic.json profiles/round6_traffic.json 2>/dev/null      # (bench.py's fallback when rocprofv3 is not usable in a run)
python bench.py --steps 20 --warmup 5 2>/dev/null | line > $O/${TAG}_bench_n1.json
# 2. the same command under the kernel tracer
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | line > $O/${TAG}_bench_under_rocprof.json
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/${TAG}_bench_kernel_stats.csv
# 3. per-kernel VALU utilisation / occupancy
bash tools/pmc_kernels.sh 1 $TAG/pmc_f32 > $O/${TAG}_pmc_f32.txt 2>&1
bash tools/pmc_kernels.sh 0 $TAG/pmc_i16 > $O/${TAG}_pmc_i16.txt 2>&1
# 4. variants
b() { name=$1; shift; python bench.py --no-cpu-baseline --no-dropin "$@" 2>/dev/null | line > $O/${TAG}_bench_$name.json; }
b graph --graph
b i16 --precision i16
b f16 --precision f16acc32
b sync_roi --sync-roi
b cycle_copy_f32 --cycle copy
b cycle_eager_f32 --cycle eager
b cycle_copy_i16 --cycle copy --precision i16
b config3_16pairs_graph --pairs 16 --graph
b config3_16pairs_streams --pairs 16
b config4_4pairs_per_gpu_streams --pairs 4
b config3_16pairs_batch_one_stream --pairs 16 --batch
b config3_16pairs_batch_3streams --pairs 16 --batch --streams 3
b config3_16pairs_one_stream --pairs 16 --streams 1
b config4_4pairs_batch_one_stream --pairs 4 --batch
b config3_16pairs_batch_graph --pairs 16 --batch --graph
b config4_4pairs_batch_graph --pairs 4 --batch --graph
b config4_forcedist_chunk --force-dist --pairs 4 --gather chunk
b config4_forcedist_single --force-dist --pairs 4 --gather single
b config4_forcedist_chunk_isx --force-dist --pairs 4 --gather chunk --gather-backend isx
b config4_forcedist_chunk_p2p --force-dist --pairs 4 --gather chunk --gather-backend p2p
b config5_ring8_8k --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2
ISX_ROLL_R23=0 python bench.py --no-cpu-baseline --no-dropin --kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 5 --warmup 2 2>/dev/null | line > $O/${TAG}_bench_config5_ring8_8k_ISX_ROLL_R23_0.json
b config5_8k_pair --kind spherical --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 10 --warmup 3
# round 4: CV_16SC3 tiles (what the reference's feed() receives) on the planned step, both arithmetic modes; the probes of the round
b s16_tiles_f32 --tile-type s16
b s16_tiles_i16 --tile-type s16 --precision i16
b steps100 --steps 100
b no_preflight --preflight-ms 0
ISX_TOP=0 python bench.py --no-cpu-baseline --no-dropin --steps 100 2>/dev/null | line > $O/${TAG}_bench_steps100_ISX_TOP0.json
ISX_TOP2=0 python bench.py --no-cpu-baseline --no-dropin --steps 100 2>/dev/null | line > $O/${TAG}_bench_steps100_ISX_TOP2_0.json
python tools/probes/ramp_probe.py > $O/${TAG}_clock_ramp.txt 2>&1
python tools/probes/fusion_probe.py > $O/${TAG}_n3_fusions.txt 2>&1
python tools/probes/graph_probe.py > $O/${TAG}_graph_vs_eager.txt 2>&1
python tools/probes/split_probe.py 1 2 > $O/${TAG}_split_strips.txt 2>&1
python tools/probes/roi_latency_probe.py > $O/${TAG}_roi_latency.txt 2>&1; ISX_ROI_POLL=0 python tools/probes/roi_latency_probe.py >> $O/${TAG}_roi_latency.txt 2>&1
python tools/probes/literal_host_probe.py 1 > $O/${TAG}_literal_host_time.txt 2>&1
for m in "1 5 literal" "0 5 literal" "1 5 sync"; do echo "== pipeline_probe $m" >> $O/${TAG}_literal_kernels.txt; python tools/pipeline_probe.py $m 2>&1 | tail -14 >> $O/${TAG}_literal_kernels.txt; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/chain_probe.hip -o /tmp/chain_probe 2>/dev/null && timeout 60 /tmp/chain_probe > $O/${TAG}_chain_by_flags.txt 2>&1
# round 5: A13 (the in-tree linear-ramp pair blend) on the record; the drop-in legs with the fused feed switched off / without narrowing; many tiles
python bench.py --a13 --steps 50 2>/dev/null | line > $O/${TAG}_bench_a13.json
for v in "ISX_FEED_FUSE=0" "ISX_FEED_NARROW=0" "ISX_FEED_STRIP=0" "ISX_FEED_FUSE=1"; do for m in "1 5 literal" "0 5 literal" "1 5 sync"; do echo "== [$v] pipeline_probe $m" >> $O/${TAG}_feed_variants.txt; env $v python tools/pipeline_probe.py $m 2>&1 | tail -14 >> $O/${TAG}_feed_variants.txt; done; done
bash tools/trace_many_tiles.sh ${TAG}_t64 --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 > $O/${TAG}_many_tiles_trace.txt 2>&1
ISX_TAB=0 python bench.py --no-cpu-baseline --no-dropin --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2 2>/dev/null | line > $O/${TAG}_bench_many_tiles_64_ISX_TAB_0.json
ISX_TAB=0 python bench.py --no-cpu-baseline --no-dropin --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2 2>/dev/null | line > $O/${TAG}_bench_many_tiles_24_ISX_TAB_0.json
b many_tiles_24 --tiles 24 --focal 9000 --yaw 0.12 --steps 6 --warmup 2
b many_tiles_64 --tiles 64 --focal 24000 --yaw 0.046 --steps 6 --warmup 2
# 4b. config 5 as ONE panorama in column strips: every rank's share at 2 / 4 / 8 ranks, each alone on this GPU (no gather)
C5="--kind spherical --tiles 8 --width 7680 --height 4320 --focal 6000 --yaw 0.275 --bands 7 --precision f16acc32 --steps 10 --warmup 3"
mkdir -p $O/strips
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
# round 6: the warp kernel's XCD-run block order (traffic and time), the level-2 fusion's bound, next step's warps under this step's blend,
# batched tile warps, config 3 as one graph with parallel chains, the host border scan against the device's
VARS="ISX_WARP_XG=0|ISX_WARP_XG=4|ISX_WARP_XG=8" bash tools/ab_env.sh > $O/${TAG}_warp_xg_ab.txt 2>&1
for g in 0 8; do ISX_WARP_XG=$g python tools/measure_traffic.py $O/${TAG}_traffic_warp_xg$g.json > /dev/null 2>&1; done
python - "$O" "$TAG" >> $O/${TAG}_warp_xg_ab.txt <<'PY'
import json, sys
O, TAG = sys.argv[1], sys.argv[2]
for g in (0, 8):
    try:
        d = json.load(open("%s/%s_traffic_warp_xg%d.json" % (O, TAG, g))); k = d.get("kernels", d)
        print("ISX_WARP_XG=%d warp_tile per launch:" % g, k.get("warp_tile"))
    except Exception as e:
        print("ISX_WARP_XG=%d: no traffic figure (%r)" % (g, e))
PY
VARS="ISX_WARP_BATCH=0|ISX_WARP_BATCH=1" bash tools/ab_env.sh > $O/${TAG}_warp_batch_ab.txt 2>&1
bash tools/probes/level2_ablation.sh run > $O/${TAG}_level2_ablation.txt 2>&1      # (needs tmp_ab/libl2_*.so: bash tools/probes/level2_ablation.sh build, here)
python tools/probes/warp_under_blend_probe.py 200 2>&1 | grep -v amdgpu.ids > $O/${TAG}_warp_under_blend.txt
b config3_16pairs_batch_graph_1chain --pairs 16 --batch --graph --streams 1
b config3_16pairs_batch_graph_4chains --pairs 16 --batch --graph --streams 4
b config3_16pairs_batch_4streams --pairs 16 --batch --streams 4
python tools/probes/roi_latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_roi_latency.txt; ISX_ROI_HOST=0 python tools/probes/roi_latency_probe.py 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_roi_latency.txt
ISX_ROI_HOST=0 python tools/probes/literal_host_probe.py 1 2>&1 | grep -v amdgpu.ids > $O/${TAG}_literal_host_time_device_roi.txt
python tools/opencv_ab.py > $O/${TAG}_opencv_ab.json 2>&1
# 5. fuzz soak
python tools/fuzz_parity.py $FUZZ 11 $O/${TAG}_fuzz_${FUZZ}s_seed11.json > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
ls $O
