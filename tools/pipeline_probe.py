"""Development probe: one 4K pair through warp -> feed -> blend on cuda:0 with per-kernel HIP-event times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import imagestitch_amd as I
from imagestitch_amd import synth, _lib

prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bands = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H, F = 3840, 2160, 3000.0
K, Rs = synth.camera_pair(W, H, F)
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
from imagestitch_amd.pipeline import PairStitcher
mode = sys.argv[3] if len(sys.argv) > 3 else "planned"
ps = PairStitcher(imgs, K, Rs, F, "cylindrical", bands, prec, 0, None, "int16", deferred="copy" if mode in ("sync", "literal") else True)
print("corners", ps.corners, "sizes", ps.sizes)
step = ps.step_sync if mode == "sync" else (ps.step_literal if mode == "literal" else ps.step)
if mode == "graph":
    ps.capture()
    step = ps.replay

for _ in range(3): out = step()
torch.cuda.synchronize()
t0 = time.time(); n = 10
for _ in range(n): out = step()
th = (time.time() - t0) / n
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print("host enqueue ms/step %.3f" % (th * 1e3))
print("ms/pair %.3f  Mpix/s %.1f" % (dt * 1e3, 2 * W * H / dt / 1e6))
lib = _lib.load()
lib.isx_profile_enable(1); lib.isx_profile_reset()
for _ in range(n): out = step()
ent = _lib.profile_entries()
tot = 0
for k, v in sorted(ent.items(), key=lambda kv: -kv[1]["ms"]):
    ms = v["ms"] / n; tot += ms
    gb = v["alg_bytes"] / n / 1e9
    print("%-16s launches/step %5.1f  ms/step %8.4f  algGB %7.4f  GB/s %8.1f" % (k, v["launches"] / n, ms, gb, gb / ms * 1e3 if ms > 0 else 0))
print("sum kernel ms/step %.4f" % tot)
