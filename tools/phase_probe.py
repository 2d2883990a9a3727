"""Where a wave of the last collapse step spends its life (needs a library built with -DISX_PHASE_TIMING, see
tools/phase_probe.sh): s_memtime ticks per phase, averaged over the waves of 10 steps of the 4K pair."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagestitch_amd import synth, _lib
from imagestitch_amd.pipeline import PairStitcher

W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
prec = {"f32": _lib.PREC_F32, "i16": _lib.PREC_I16, "f16acc32": _lib.PREC_F16ACC32}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, None, "int16")
lib = _lib.load()
if not hasattr(lib, "isx_debug_phase"):
    sys.exit("libimagestitch_hip.so was built without -DISX_PHASE_TIMING (tools/phase_probe.sh builds it)")
lib.isx_debug_phase.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for _ in range(3):
    ps.step()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 12)()
lib.isx_debug_phase(buf, 1)
n = 10
for _ in range(n):
    ps.step()
torch.cuda.synchronize()
lib.isx_debug_phase(buf, 0)
v = list(buf)
waves = v[11]
if os.environ.get("ISX_ROLL", "1") != "0":
    names = ["block order, tile selection", "slot descriptors, lane offsets", "loads issued", "all loads waited for (timing build only)",
             "tiles: row / column filters, decode, accumulate", "out_1: filters, normalise, convert, stores issued"]
    tot = sum(v[:11])
    print("k_collapse_roll<%s, U8> on a 4K pair:" % (sys.argv[1] if len(sys.argv) > 1 else "f32"), "%.0f waves per launch, wave lifetime %.0f s_memtime ticks" % (waves / n, tot / waves))
    for k in range(6):
        print("  %-58s %8.1f ticks/wave  %5.1f %%" % (names[k], v[k] / waves, 100.0 * v[k] / tot))
    sys.exit(0)
names = ["round 0: tile descriptors (scalar loads)", "round 0: coarse tiles issued (LDS-DMA)", "round 0: fine pixels issued", "round 0: memory + barrier wait",
         "round 1: tile descriptors", "round 1: coarse tiles + out issued", "round 1: fine pixels issued", "round 1: memory + barrier wait",
         "round 0: decode + pyrUp + accumulate", "round 1: decode + pyrUp + accumulate", "epilogue: normalise, pyrUp(out), convert, stores issued"]
tot = sum(v[:11])
print("k_collapse_gather<%s, U8, FINE0> on a 4K pair:" % (sys.argv[1] if len(sys.argv) > 1 else "f32"), "%.0f waves per launch, wave lifetime %.0f s_memtime ticks" % (waves / n, tot / waves))
for k in (0, 1, 2, 3, 8, 4, 5, 6, 7, 9, 10):
    print("  %-58s %8.1f ticks/wave  %5.1f %%" % (names[k], v[k] / waves, 100.0 * v[k] / tot))
