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
warper = I.CylindricalWarper().create(F)
mb = I.MultiBandBlender(False, bands, prec)
corners, wimgs, wmasks = [], [], []
for i in range(2):
    c, wi, wm = warper.warp_with_mask(imgs[i], K, Rs[i])
    corners.append(c); wimgs.append(wi); wmasks.append(wm)
sizes = [(m.shape[1], m.shape[0]) for m in wmasks]
print("corners", corners, "sizes", sizes)
seam = synth.seam_masks(corners, [m.cpu().numpy() for m in wmasks])
seam = [torch.from_numpy(s).to(dev) for s in seam]

def step():
    cs = []
    for i in range(2):
        c, wi, wm = warper.warp_with_mask(imgs[i], K, Rs[i], dst_img=wimgs[i], dst_mask=wmasks[i])
        cs.append(c)
    mb.prepare(cs, sizes)
    for i in range(2):
        mb.feed_u8(wimgs[i], seam[i], cs[i])
    return mb.blend(out_f32=(prec != 0))

for _ in range(3): out = step()
torch.cuda.synchronize()
t0 = time.time(); n = 10
for _ in range(n): out = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
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
