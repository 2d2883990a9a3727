#!/usr/bin/env bash
# Round 6: the occupancy / workgroup knobs of the two biggest pyramid launches once more, on today's kernels (their settings date from rounds 3 - 5,
# before the level-1 shorts and the hoisted loads): k_collapse_roll as 2-wave workgroups, at 4 / 6 waves per SIMD; the level-0 pyrDown at 4 / 8.
#   bash tools/probes/retune_round6.sh build (here)      gpurun -- 'bash tools/probes/retune_round6.sh run'
cd "$(dirname "$0")/../.."
if [ "${1:-run}" = build ]; then
  bash tools/build_variant.sh rt_base blend.hip "" & bash tools/build_variant.sh rt_w2 blend.hip "-DROLL_WAVES_N=2" &
  bash tools/build_variant.sh rt_wpe4 blend.hip "-DROLL_WPE2=4" & bash tools/build_variant.sh rt_wpe6 blend.hip "-DROLL_WPE2=6" & wait
  bash tools/build_variant.sh rt_pd4 blend.hip "-DISX_PD0_WPE=4" & bash tools/build_variant.sh rt_pd8 blend.hip "-DISX_PD0_WPE=8" & wait
else
  VARS="rt_base rt_w2 rt_wpe4 rt_wpe6 rt_pd4 rt_pd8" REPS=3 bash tools/ab_libs.sh --steps 100 --warmup 10
fi
