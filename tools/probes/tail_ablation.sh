#!/usr/bin/env bash
# what could ONE launch for the levels >= 2 (the three small pyrDown steps + k_collapse_top) give at most?  Timing-only builds
# of blend.hip (wrong pixels): TAIL1 = the three pyrDown launches as one block per tile (their launch + dependency stay, their work goes),
# TAIL3 = k_collapse_top as one block as well, TAIL7 = the four launches not issued at all (the bound of ANY fusion: they cost nothing).
# Build here:  bash tools/probes/tail_ablation.sh build      Run on the GPU box:  gpurun -- 'bash tools/probes/tail_ablation.sh run'
cd "$(dirname "$0")/../.."
if [ "${1:-run}" = build ]; then
  bash tools/build_variant.sh tail0 blend.hip "" & bash tools/build_variant.sh tail1 blend.hip "-DISX_TAIL_ABL=1" &
  bash tools/build_variant.sh tail3 blend.hip "-DISX_TAIL_ABL=3" & bash tools/build_variant.sh tail7 blend.hip "-DISX_TAIL_ABL=7" & wait
else
  VARS="tail0 tail1 tail3 tail7" REPS=3 bash tools/ab_libs.sh --steps 100 --warmup 10
fi
