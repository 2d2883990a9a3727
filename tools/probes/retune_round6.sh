#!/usr/bin/env bash
# Round 6: the occupancy / workgroup knobs of the two biggest pyramid launches once more, on today's kernels (their settings date from rounds 3 - 5,
# before the level-1 shorts and the hoisted loads): k_collapse_roll as 2-wave workgroups, at 4 / 6 waves per SIMD; the level-0 pyrDown at 4 / 8.
#   bash tools/probes/retune_round6.sh build (here)      gpurun -- 'bash tools/probes/retune_round6.sh run'
cd "$(dirname "$0")/../.."
if [ "${1:-run}" = build ]; then
  bash tools/build_variant.sh rt_base blend.hip "" & bash tools/build_variant.sh rt_w2 blend.hip "-DROLL_WAVES_N=2" &
  bash tools/build_variant.sh rt_wpe4 blend.hip "-DROLL_WPE2=4" & bash tools/build_variant.sh rt_wpe6 blend.hip "-DROLL_WPE2=6" & wait
  bash tools/build_variant.sh rt_pd4 blend.hip "-DISX_PD0_WPE=4" & bash tools/build_variant.sh rt_pd8 blend.hip "-DISX_PD0_WPE=8" & wait
elif [ "$1" = build2 ]; then      # second sweep: the level-1 step's and the top launch's occupancy, the level-0 pyrDown's XCD band order, the warp kernel's block shape
  bash tools/build_variant.sh rt2_base blend.hip "" & bash tools/build_variant.sh rt2_g5 blend.hip "-DISX_GATHER_LEVEL_WPE=5" &
  bash tools/build_variant.sh rt2_t5 blend.hip "-DISX_TOP2_WPE=5" & bash tools/build_variant.sh rt2_band0 blend.hip "-DISX_PD0_BAND=0" & wait
  bash tools/build_variant.sh rt2_ww2 warp.hip "-DWARP_WAVES=2" & bash tools/build_variant.sh rt2_ww8 warp.hip "-DWARP_WAVES=8" & bash tools/build_variant.sh rt2_wwpe8 warp.hip "-DWARP_WPE=8" & wait
elif [ "$1" = run2 ]; then
  VARS="rt2_base rt2_g5 rt2_t5 rt2_band0 rt2_ww2 rt2_ww8 rt2_wwpe8" REPS=3 bash tools/ab_libs.sh --steps 100 --warmup 10
else
  VARS="rt_base rt_w2 rt_wpe4 rt_wpe6 rt_pd4 rt_pd8" REPS=3 bash tools/ab_libs.sh --steps 100 --warmup 10
fi
