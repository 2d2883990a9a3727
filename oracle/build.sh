#!/usr/bin/env bash
# oracle/build.sh — builds the TEST-ONLY checker libraries.
#   oracle/liboracle.so        : the C restatement (oracle.c)
#   oracle/_ref/libref_warp.so : W:30-63 of the reference compiled VERBATIM from where it lies
#                                under /root/reference (only when that tree is present, i.e. in
#                                the build container; the GPU box uses the prebuilt file).
# The rest of the reference's hot path (remap, pyramids, MultiBandBlender) needs OpenCV 3.4.2
# headers and libraries that this image lacks: unbuildable here, see DESIGN.md.
set -euo pipefail
cd "$(dirname "$0")"
CC=${CC:-gcc}
CXX=${CXX:-g++}
# -ffp-contract=off: the fp32 association in oracle.c is the spec; never let gcc fuse a*b+c.
$CC -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -Wall -Wextra -o liboracle.so oracle.c -lm
echo "built oracle/liboracle.so"
REF="/root/reference/圆柱面投影变换/圆柱面投影变换/圆柱面投影.cpp"
if [ -f "$REF" ]; then
    mkdir -p _ref
    # lines 30-63, GB18030 -> UTF-8 (they are pure ASCII in that range), straight into the compiler
    { echo '#include <cmath>'; echo '#include <limits>';
      sed -n '30,63p' "$REF" | iconv -f GB18030 -t UTF-8;
      cat ref_shim.inc; } | $CXX -O2 -ffp-contract=off -fPIC -shared -x c++ - -o _ref/libref_warp.so
    echo "built oracle/_ref/libref_warp.so from $REF:30-63"
else
    echo "reference tree absent: keeping prebuilt oracle/_ref (if any)"
fi
