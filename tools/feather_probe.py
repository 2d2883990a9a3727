"""Development probe: the demos' default blend (dilate+AND, FeatherBlender, W:278-313) on a 4K-sized pair of device
tiles, per-kernel HIP-event times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagestitch_amd as I
from imagestitch_amd import _lib

dev = torch.device("cuda:0")
sizes, corners = [(3417, 2169), (3417, 2169)], [(-1709, -1085), (459, -1085)]
g = torch.Generator(device=dev); g.manual_seed(3)
imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device=dev, generator=g) for (w, h) in sizes]
masks = [torch.full((h, w), 255, dtype=torch.uint8, device=dev) for (w, h) in sizes]
for m in masks: m[:40, :300] = 0
fb = I.FeatherBlender(False, 0.1)
fb.set_deferred_level0(len(sys.argv) < 2 or sys.argv[1] != "eager")

def step():
    fb.prepare(corners, sizes)
    for im, m, c in zip(imgs, masks, corners):
        mm = I.dilate_and(m, 20, 20, other=m)
        fb.feed_u8(im, mm, c) if hasattr(fb, "feed_u8") else fb.feed(im.to(torch.int16), mm, c)
    return fb.blend()

for _ in range(2): step()
torch.cuda.synchronize()
n = 5
t0 = time.time()
for _ in range(n): step()
torch.cuda.synchronize()
print("ms/pair (incl. output allocation) %.3f" % ((time.time() - t0) / n * 1e3))
lib = _lib.load()
lib.isx_profile_enable(1); lib.isx_profile_reset()
for _ in range(n): step()
ent = _lib.profile_entries()
for k, v in sorted(ent.items(), key=lambda kv: -kv[1]["ms"]):
    print("%-16s launches/step %5.1f  ms/step %8.4f" % (k, v["launches"] / n, v["ms"] / n))
