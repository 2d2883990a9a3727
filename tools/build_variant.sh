#!/usr/bin/env bash
# tools/build_variant.sh <NAME> <file.hip> "<extra -D flags>": builds tmp_ab/lib<NAME>.so = the in-tree objects with <file.hip> recompiled under the
# extra flags (same-box A/B runs on the GPU box: tools/ab_libs.sh, tools/ab_literal.sh).  tmp_ab/ is scratch (git-ignored, travels with gpurun).
set -euo pipefail
name=$1; src=$2; extra=$3
cd "$(dirname "$0")/../imagestitch_amd/csrc"
mkdir -p ../../tmp_ab/obj_$name
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function $extra"
/opt/rocm/bin/hipcc $FLAGS -x hip -c "$src" -o ../../tmp_ab/obj_$name/${src%.*}.o
objs=""
for o in build/*.o; do
  b=$(basename $o)
  if [ "$b" = "${src%.*}.o" ]; then objs="$objs ../../tmp_ab/obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tmp_ab/lib$name.so $objs -ldl -Wl,-rpath,/opt/rocm/lib
echo "built tmp_ab/lib$name.so"
