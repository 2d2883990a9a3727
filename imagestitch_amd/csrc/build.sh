#!/usr/bin/env bash
# Builds imagestitch_amd/csrc/libimagestitch_hip.so for gfx950 (MI355X) with hipcc.
#   -ffp-contract=off : the fp32 kernels must evaluate a*b+c as two rounded operations, exactly
#                       like the (non-FMA) CPU reference code they are bit-compared against.
#   -fno-slp-vectorize : the SLP vectoriser pairs the 3-channel fp32 arithmetic of the pyramid kernels into
#                       v_pk_mul/add_f32, but the records are 3 floats wide, so every pair costs v_mov's to line its
#                       halves up in an even register pair (and registers: 128 VGPRs + scratch against 118 without).
#                       Measured (rocprofv3 and same-box A/B runs): the last collapse step 90 us with it, 82 us without;
#                       the other pyramid kernels 3-7 % less; the warp kernel 2-3 % less.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function ${ISX_EXTRA_FLAGS:-}"
mkdir -p build
pids=()
for f in isx_core.cpp imgio.cpp jpegdec.cpp seamfind.cpp gather.cpp warp.hip blend.hip prep.hip linear_blend.hip seam.hip; do
    [ -f "$f" ] || continue
    o=build/${f%.*}.o
    if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ isx_internal.hpp -nt "$o" ] || [ isx_device.hpp -nt "$o" ] || [ collapse_roll.inc -nt "$o" ] || [ collapse_top2.inc -nt "$o" ] || [ pyrdown_l0.inc -nt "$o" ] || [ ../../include/imagestitch_hip.h -nt "$o" ] || [ build.sh -nt "$o" ] || [ collapse_top.inc -nt "$o" ]; then
        $HIPCC $FLAGS -x hip -c "$f" -o "$o" &
        pids+=($!)
    fi
done
# host-only code (no HIP): detectResultRoi's border scan on the caller's thread, AVX2 behind a run-time CPU check (function target attributes)
if [ ! -f build/roihost.o ] || [ roihost.cpp -nt build/roihost.o ] || [ build.sh -nt build/roihost.o ]; then
    ${CXX:-g++} -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -c roihost.cpp -o build/roihost.o &
    pids+=($!)
fi
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libimagestitch_hip.so build/*.o -ldl -Wl,-rpath,/opt/rocm/lib
echo "built $(pwd)/libimagestitch_hip.so"
