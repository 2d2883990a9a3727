"""BASELINE config 5 shape: spherical warp + 7-band F16ACC32 blend (8K tiles in the config; here one
1/4-scale pair against the oracle bit-exactly, plus one full 8K pair through size-independent properties)."""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu


def test_spherical_7band_f16acc32_against_oracle(gpu, oracle):
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 1920, 1080, 1500.0
    K, Rs = synth.camera_pair(W, H, F, yaw=0.275)          # yaw step 0.55 rad (SURVEY config 5)
    imgs = [synth.make_tile(H, W, 50 + i) for i in range(2)]
    dev = torch.device("cuda:0")
    ps = PairStitcher([torch.from_numpy(i).to(dev) for i in imgs], K, Rs, F, "spherical", 7, gpu.PREC_F16ACC32, 0, None, "int16")
    out, omask = [t.cpu().numpy() for t in ps.step_sync()]
    o_w, o_m = [], []
    for i in range(2):
        c, wi, _ = oracle.warp_u8(oracle.SPH, F, K, Rs[i], imgs[i], 1, 2)
        _, wm, _ = oracle.warp_u8(oracle.SPH, F, K, Rs[i], np.full((H, W), 255, np.uint8), 0, 0)
        assert c == ps.corners[i]
        assert np.array_equal(ps.warped[i].cpu().numpy(), wi) and np.array_equal(ps.wmasks[i].cpu().numpy(), wm)
        o_w.append(wi); o_m.append(wm)
    seam = [s.cpu().numpy() for s in ps.seam]
    ob = oracle.MultiBand(7, oracle.F16ACC32)
    ob.prepare(ps.corners, ps.sizes)
    assert ob.num_bands == 7
    for i in range(2):
        ob.feed(o_w[i].astype(np.int16), seam[i], ps.corners[i])
    od, om = ob.blend(False)
    assert np.array_equal(omask, om) and np.array_equal(out, od)


def test_full_size_8k_spherical_pair_properties(gpu):
    """2 x 7680x4320 tiles, spherical f=6000, 7 bands, F16ACC32: sizes, mask = union of the seam masks,
    zero outside, and the blend of the pair stays within the input range where both tiles agree."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 7680, 4320, 6000.0
    K, Rs = synth.camera_pair(W, H, F, yaw=0.275)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(5)
    imgs = [torch.randint(100, 140, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(2)]
    ps = PairStitcher(imgs, K, Rs, F, "spherical", 7, gpu.PREC_F16ACC32, 0, None, "int16")
    out, omask = ps.step_sync()
    torch.cuda.synchronize()
    x0 = min(c[0] for c in ps.corners); y0 = min(c[1] for c in ps.corners)
    union = torch.zeros_like(omask)
    for i in range(2):
        cx, cy = ps.corners[i][0] - x0, ps.corners[i][1] - y0
        h, w = ps.seam[i].shape
        union[cy:cy + h, cx:cx + w] |= ps.seam[i]
    assert torch.equal(omask, union)
    assert int(out[omask == 0].abs().max()) == 0
    inside = out[omask == 255].float()
    assert 90 <= float(inside.min()) and float(inside.max()) <= 150        # inputs are U{100..139}
    assert abs(float(inside.mean()) - 119.5) < 1.5                         # -1 bias of the saturate_cast path stays small
    del ps
    torch.cuda.empty_cache()
