"""One pair of very large tiles through the planned step: a checksum of mosaic and mask, to compare kernel variants (environment switches
ISX_ROLL / ISX_PD0, read when the library is first used) at sizes the oracle cannot reach.
    python tools/probes/big_pair_probe.py [width height focal [int16|uint8|float32]]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagestitch_amd import synth, _lib  # noqa: E402
from imagestitch_amd.pipeline import PairStitcher  # noqa: E402

W, H, F = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (15360, 8640, 12000.0)
ODT = sys.argv[4] if len(sys.argv) > 4 else "int16"
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev); gen.manual_seed(7)
K, Rs = synth.camera_pair(W, H, F)
imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(2)]
for prec, name in ((_lib.PREC_F32, "f32"), (_lib.PREC_I16, "i16")):
    p = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, None, ODT)
    out, m = p.step()
    torch.cuda.synchronize()
    p.check_plan()
    a = out.to(torch.int64)
    w = torch.arange(1, a.shape[1] + 1, device=dev, dtype=torch.int64).view(1, -1, 1)
    print(name, tuple(out.shape), int((a * w).sum().item()), int(a.abs().sum().item()), int(m.to(torch.int64).sum().item()))
    del p, out, m, a
    torch.cuda.empty_cache()
