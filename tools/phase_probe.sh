#!/usr/bin/env bash
# Builds the library with the phase stamps of the last collapse step compiled in, runs tools/phase_probe.py on the GPU box and
# rebuilds the product library.  Run from the repo root in the build container:  bash tools/phase_probe.sh
set -euo pipefail
cd "$(dirname "$0")/.."
touch imagestitch_amd/csrc/blend.hip
ISX_EXTRA_FLAGS=-DISX_PHASE_TIMING bash imagestitch_amd/csrc/build.sh
/usr/local/graft/bin/gpurun --timeout 600 -- 'mkdir -p gpurun_out; python tools/phase_probe.py 2>/dev/null | tee gpurun_out/phase_probe.txt' || true
touch imagestitch_amd/csrc/blend.hip
bash imagestitch_amd/csrc/build.sh
