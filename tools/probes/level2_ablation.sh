#!/usr/bin/env bash
# "let the level-0 pyrDown emit level 2 as well" - what could that give at most?  Timing-only builds of blend.hip (wrong pixels):
#   l2_0  = the tree;
#   l2_8  = the level 1 -> 2 pyrDown launch not issued (what the fusion removes: the launch and its re-read of level 1);
#   l2_16 = the level-0 pyrDown doing 1.3 x its work (a grid 1.3 x as tall redoing rows: the halo of level 1 a block must recompute to own a patch of
#           level 2 - 19 x 62 level-1 pixels for the 16 x 58 it owns is 1.27 x, 41 level-0 rows for 32 is 1.28 x; the second pass itself - another
#           barrier, a column pass over LDS, the level-2 stores - is NOT in this build);
#   l2_24 = both: the bound of the fusion before its own second phase costs anything.
# Build here:  bash tools/probes/level2_ablation.sh build      Run on the GPU box:  gpurun -- 'bash tools/probes/level2_ablation.sh run'
cd "$(dirname "$0")/../.."
if [ "${1:-run}" = build ]; then
  bash tools/build_variant.sh l2_0 blend.hip "" & bash tools/build_variant.sh l2_8 blend.hip "-DISX_TAIL_ABL=8" &
  bash tools/build_variant.sh l2_16 blend.hip "-DISX_TAIL_ABL=16" & bash tools/build_variant.sh l2_24 blend.hip "-DISX_TAIL_ABL=24" & wait
else
  VARS="l2_0 l2_8 l2_16 l2_24" REPS=3 bash tools/ab_libs.sh --steps 100 --warmup 10
fi
