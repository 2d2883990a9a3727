"""The BASELINE.json configurations as stated, each as a parity test through the C-ABI:

  config 1: 2 x 1920x1080 tiles, cylindrical warp + 3-band blend                       (oracle, bit-exact)
  config 3: 16 independent 4K pairs resident on one GPU, one hipGraph per pair           (graphs == eager == oracle)
  config 5: 8 tiles in ONE mosaic, spherical warp, yaw step 0.55 rad, 7 bands, F16ACC32  (1/4 scale: oracle, bit-exact;
            full 8K: size-independent properties)

Config 2 (one 4K pair, 5 bands, fp32) is tests/test_gpu_blend.py::test_full_size_4k_pair_*; config 4 is config 3's pairs
sharded over ranks (tests/test_dist_gloo.py, tests/test_gpu_dist.py).
"""
import numpy as np
import pytest

from imagestitch_amd import synth

pytestmark = pytest.mark.gpu


def _oracle_mosaic(oracle, kind, F, K, Rs, imgs, bands, prec, out_f32):
    """The reference's call sequence on the CPU oracle: warp(image) + warp(mask) per tile (W:229,232), seam stand-in,
    prepare / feed x n / blend (W:281,294,302,313)."""
    H, W = imgs[0].shape[:2]
    corners, warped, wmasks = [], [], []
    for im, R in zip(imgs, Rs):
        c, wi, _ = oracle.warp_u8(kind, F, K, R, im, oracle.LINEAR, oracle.BORDER_REFLECT)
        _, wm, _ = oracle.warp_u8(kind, F, K, R, np.full((H, W), 255, np.uint8), oracle.NEAREST, oracle.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    seam = synth.seam_masks(corners, wmasks)
    sizes = [(m.shape[1], m.shape[0]) for m in wmasks]
    ob = oracle.MultiBand(bands, prec)
    ob.prepare(corners, sizes)
    for wi, sm, c in zip(warped, seam, corners):
        ob.feed(wi.astype(np.int16), sm, c)
    od, om = ob.blend(out_f32)
    return corners, warped, wmasks, seam, od, om


def test_config1_two_1080p_tiles_3_bands(gpu, oracle):
    """BASELINE config 1 on the GPU path: the sizes and the band count the reference's own demo would run."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 1920, 1080, 1500.0
    K, Rs = synth.camera_pair(W, H, F)
    imgs = [synth.make_tile(H, W, 10 + i) for i in range(2)]
    dev = torch.device("cuda:0")
    corners, warped, wmasks, seam, od, om = _oracle_mosaic(oracle, oracle.CYL, F, K, Rs, imgs, 3, oracle.I16, False)
    for deferred in (False, "copy", True):
        ps = PairStitcher([torch.from_numpy(i).to(dev) for i in imgs], K, Rs, F, "cylindrical", 3, gpu.PREC_I16, 0, None, "int16", deferred=deferred)
        for out, omask in (ps.step_sync(), ps.step()):
            assert ps.corners == corners
            assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(ps.warped, warped))
            assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(ps.wmasks, wmasks))
            assert np.array_equal(omask.cpu().numpy(), om) and np.array_equal(out.cpu().numpy(), od)
        ps.check_plan()


def test_config3_sixteen_4k_pairs_eager_on_four_streams(gpu, oracle):
    """BASELINE config 3 as bench.py runs it by default (--pairs 16: eager launches, the pairs spread over 4 streams so that one pair's
    small pyramid levels run under another's large kernels): three steps of 16 concurrent pairs; every mosaic equals the same pair's
    mosaic from a serial single-stream run, and one pair equals the oracle."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F, NP, NS = 3840, 2160, 3000.0, 16, 4
    K, Rs = synth.camera_pair(W, H, F)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    check_pair = 5
    host_imgs = [synth.make_tile(H, W, 320 + i) for i in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
    inputs = []
    for p in range(NP):
        if p == check_pair:
            inputs.append([torch.from_numpy(i).to(dev) for i in host_imgs])
        else:
            gen.manual_seed(synth.SEED0 + 50 + p)
            inputs.append([torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(2)])
    serial = []
    for p in range(NP):                                    # the expected mosaics: one stitcher, one stream, one pair after the other
        ps = PairStitcher(inputs[p], K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "float32")
        out, omask = ps.step()
        serial.append((out.clone(), omask.clone()))
        ps.check_plan()
        del ps
    torch.cuda.synchronize()
    pairs = [PairStitcher(inputs[p], K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, streams[p % NS], "float32") for p in range(NP)]
    torch.cuda.synchronize()
    for rep in range(3):
        for ps in pairs:
            ps.out.zero_(); ps.out_mask.zero_()
        torch.cuda.synchronize()
        for p, ps in enumerate(pairs):                     # 16 steps enqueued back to back over the 4 streams
            with torch.cuda.stream(streams[p % NS]):
                ps.step()
        torch.cuda.synchronize()
        for p, ps in enumerate(pairs):
            assert torch.equal(ps.out, serial[p][0]) and torch.equal(ps.out_mask, serial[p][1]), "pair %d, step %d" % (p, rep)
    for ps in pairs:
        ps.check_plan()
    _, _, _, _, od, om = _oracle_mosaic(oracle, oracle.CYL, F, K, Rs, host_imgs, 5, oracle.F32, True)
    assert np.array_equal(pairs[check_pair].out_mask.cpu().numpy(), om)
    assert np.array_equal(pairs[check_pair].out.cpu().numpy(), od)
    del pairs, serial, inputs
    torch.cuda.empty_cache()


def test_config3_sixteen_4k_pairs_as_graphs(gpu, oracle):
    """BASELINE config 3: 16 independent 4K pairs resident on one GPU, each captured as its own hipGraph, replayed twice;
    every replayed mosaic equals the eager (launch by launch) run of the same pair, and one pair equals the oracle."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F, NP = 3840, 2160, 3000.0, 16
    K, Rs = synth.camera_pair(W, H, F)
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    check_pair = 11                                        # this one gets numpy-made tiles so that the oracle can run it
    host_imgs = [synth.make_tile(H, W, 300 + i) for i in range(2)]
    pairs = []
    for p in range(NP):
        if p == check_pair:
            imgs = [torch.from_numpy(i).to(dev) for i in host_imgs]
        else:
            gen.manual_seed(synth.SEED0 + p)
            imgs = [torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen) for _ in range(2)]
        pairs.append(PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "float32"))
    eager = []
    for ps in pairs:                                       # serial eager run: the expected mosaics
        out, omask = ps.step()
        eager.append((out.clone(), omask.clone()))
    torch.cuda.synchronize()
    for ps in pairs:
        ps.capture()
    for rep in range(2):
        for ps in pairs:                                   # overwrite the outputs so that a replay that did nothing is caught
            ps.out.zero_(); ps.out_mask.zero_()
        torch.cuda.synchronize()
        for ps in pairs:                                   # 16 graphs in flight on 16 streams
            ps.replay()
        torch.cuda.synchronize()
        for p, ps in enumerate(pairs):
            assert torch.equal(ps.out, eager[p][0]) and torch.equal(ps.out_mask, eager[p][1]), "pair %d, replay %d" % (p, rep)
    for ps in pairs:
        ps.check_plan()
    # distinct inputs gave distinct mosaics (the 16 graphs did not alias each other's buffers)
    assert not torch.equal(pairs[0].out, pairs[1].out)
    _, _, _, _, od, om = _oracle_mosaic(oracle, oracle.CYL, F, K, Rs, host_imgs, 5, oracle.F32, True)
    assert np.array_equal(pairs[check_pair].out_mask.cpu().numpy(), om)
    assert np.array_equal(pairs[check_pair].out.cpu().numpy(), od)      # 0 ULP (north_star allows 1)
    del pairs, eager
    torch.cuda.empty_cache()


def test_config5_ring_of_8_tiles_quarter_scale_against_oracle(gpu, oracle):
    """BASELINE config 5 at 1/4 scale: 8 x 1920x1080 tiles in ONE mosaic, spherical warp f = 1500, yaw step 0.55 rad,
    7 bands, fp16 pyramid with fp32 accumulate — through the deferred cycle (8 of the at most 20 tiles it holds), through the
    copying deferred cycle and through the eager cycle; warps, masks and the blended mosaic bit-exact vs the oracle."""
    import torch
    from imagestitch_amd.pipeline import MosaicStitcher
    W, H, F, N = 1920, 1080, 1500.0, 8
    K, Rs = synth.camera_ring(W, H, F, N, 0.55)
    imgs = [synth.make_tile(H, W, 500 + i) for i in range(N)]
    corners, warped, wmasks, seam, od, om = _oracle_mosaic(oracle, oracle.SPH, F, K, Rs, imgs, 7, oracle.F16ACC32, False)
    assert od.shape[1] > 6 * W // 2                        # one wide mosaic, not eight separate ones
    dev = torch.device("cuda:0")
    for deferred in (True, "copy", False):
        ms = MosaicStitcher([torch.from_numpy(i).to(dev) for i in imgs], K, Rs, F, "spherical", 7, gpu.PREC_F16ACC32, 0, None, "int16", deferred=deferred)
        assert ms.corners == corners and ms.L == 7
        out, omask = ms.step_sync()
        for i in range(N):
            assert np.array_equal(ms.warped[i].cpu().numpy(), warped[i]) and np.array_equal(ms.wmasks[i].cpu().numpy(), wmasks[i]), i
            assert np.array_equal(ms.seam[i].cpu().numpy(), seam[i])
        assert np.array_equal(omask.cpu().numpy(), om), deferred
        assert np.array_equal(out.cpu().numpy(), od), deferred
        del ms
    torch.cuda.empty_cache()


def test_config5_ring_of_8_tiles_full_8k_properties(gpu):
    """BASELINE config 5 as stated: 8 x 7680x4320 tiles, spherical f = 6000, yaw step 0.55 rad, 7 bands, F16ACC32, one
    mosaic (about 30 000 x 4 700 pixels).  Too large for the oracle in test time, so: sizes, result mask = union of the
    seam masks, zeros outside it, values inside the range of the inputs, and the mosaic of 8 copies of tile content that
    is constant per channel reproduces that constant (blend of identical values, up to the cast bias)."""
    import torch
    from imagestitch_amd.pipeline import MosaicStitcher
    W, H, F, N = 7680, 4320, 6000.0, 8
    K, Rs = synth.camera_ring(W, H, F, N, 0.55)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(55)
    imgs = [torch.randint(100, 140, (H, W, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(N)]
    ms = MosaicStitcher(imgs, K, Rs, F, "spherical", 7, gpu.PREC_F16ACC32, 0, None, "int16")
    assert ms.L == 7
    out, omask = ms.step_sync()
    torch.cuda.synchronize()
    x0 = min(c[0] for c in ms.corners); y0 = min(c[1] for c in ms.corners)
    assert out.shape[1] == max(c[0] + s[0] for c, s in zip(ms.corners, ms.sizes)) - x0 and out.shape[1] > 25000
    union = torch.zeros_like(omask)
    for i in range(N):
        cx, cy = ms.corners[i][0] - x0, ms.corners[i][1] - y0
        h, w = ms.seam[i].shape
        union[cy:cy + h, cx:cx + w] |= ms.seam[i]
    assert torch.equal(omask, union)
    assert int(out[omask == 0].abs().max()) == 0
    inside = out[omask == 255].float()
    assert 85 <= float(inside.min()) and float(inside.max()) <= 155        # inputs are U{100..139}; Laplacian overshoot is bounded
    assert abs(float(inside.mean()) - 119.5) < 1.5
    first = out.clone()
    out2, _ = ms.step_sync()                                               # idempotent: same inputs, same bits
    assert torch.equal(first, out2)
    del first
    # constant tiles: every band but the top is zero inside the tiles, the blend returns the constant
    for im in imgs:
        im[:, :, 0] = 60; im[:, :, 1] = 128; im[:, :, 2] = 200
    out3, omask3 = ms.step_sync()
    torch.cuda.synchronize()
    assert torch.equal(omask3, union)
    core = torch.zeros_like(omask3, dtype=torch.bool)
    core[out3.shape[0] // 2 - 600: out3.shape[0] // 2 + 600, 2000:-2000] = True    # away from the mosaic's outer border
    core &= omask3 == 255
    for c, v in enumerate((60, 128, 200)):
        vals = out3[:, :, c][core]
        assert int(vals.min()) >= v - 1 and int(vals.max()) <= v + 1, (c, int(vals.min()), int(vals.max()))
    del ms, imgs
    torch.cuda.empty_cache()


def test_config5_planned_step_and_graph_equal_the_synchronous_step(gpu, oracle):
    """The spherical projector on the sync-free path (round 3: its ROI is a border scan on the device): the planned step (ROI verified on
    the device, no host round trip), its hipGraph replay and the synchronous step (every corner returned to the host) produce the same
    mosaic, which is the oracle's; a plan that does not fit the cameras raises the sticky mismatch flag."""
    import torch
    from imagestitch_amd import IsxError
    from imagestitch_amd.pipeline import MosaicStitcher
    W, H, F, N = 960, 540, 750.0, 4
    K, Rs = synth.camera_ring(W, H, F, N, 0.55)
    imgs = [synth.make_tile(H, W, 700 + i) for i in range(N)]
    corners, warped, wmasks, seam, od, om = _oracle_mosaic(oracle, oracle.SPH, F, K, Rs, imgs, 5, oracle.F32, False)
    dev = torch.device("cuda:0")
    d_imgs = [torch.from_numpy(i).to(dev) for i in imgs]
    ms = MosaicStitcher(d_imgs, K, Rs, F, "spherical", 5, gpu.PREC_F32, 0, None, "int16")
    assert ms.corners == corners
    out_sync = ms.step_sync()[0].clone()
    assert np.array_equal(out_sync.cpu().numpy(), od)
    out_planned = ms.step()[0].clone()
    ms.check_plan()                                     # the device-side check agreed with the plan
    assert torch.equal(out_planned, out_sync)
    ms.capture()
    ms.out.zero_()
    out_graph = ms.replay()[0]
    torch.cuda.synchronize()
    ms.check_plan()
    assert torch.equal(out_graph, out_sync)
    # a stale plan: the same ROIs with cameras that look elsewhere
    K2, Rs2 = synth.camera_ring(W, H, F, N, 0.40)
    stale = MosaicStitcher(d_imgs, K, Rs, F, "spherical", 5, gpu.PREC_F32, 0, None, "int16")
    stale.Rs = Rs2
    stale.step()
    with pytest.raises(IsxError):
        stale.check_plan()
    del ms, stale


def test_spherical_roi_device_border_scan_equals_host_scan(gpu, oracle):
    """detectResultRoi of the spherical projector (device border scan + host refinement + pole tests) against the oracle's host-only scan:
    random cameras incl. ones that look at a pole (pitch near +-90 degrees), several source sizes."""
    import imagestitch_amd as I
    rng = np.random.RandomState(5)
    wp = I.SphericalWarper(0)
    n_pole = 0
    for trial in range(60):
        W, H = int(rng.randint(40, 900)), int(rng.randint(40, 700))
        f = float(rng.uniform(0.4, 2.0) * max(W, H))
        K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
        pitch = rng.uniform(-1.7, 1.7) if trial % 3 == 0 else rng.uniform(-0.4, 0.4)
        yaw, roll = rng.uniform(-3.0, 3.0), rng.uniform(-0.3, 0.3)
        cx, sx, cy, sy, cz, sz = np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw), np.cos(roll), np.sin(roll)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]); Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]); Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = (Ry @ Rx @ Rz).astype(np.float32)
        scale = float(rng.uniform(0.5, 1.5) * f)
        w = wp.create(scale)
        got = w.warpRoi((W, H), K, R)
        exp, _ = oracle.detect_roi(oracle.SPH, scale, K, R, W, H)
        assert tuple(int(v) for v in got) == tuple(int(v) for v in exp), (trial, W, H, f, pitch, yaw, roll, got, exp)
        n_pole += abs(pitch) > 1.2
    assert n_pole >= 5


@pytest.mark.parametrize("prec_name", ["f32", "i16", "f16acc32"])
def test_batched_blend_equals_the_serial_blends(gpu, prec_name):
    """isx_blender_blend_batch (round 3: ONE chain of launches for several mosaics, the mosaic as grid.z): 8 pairs with different images
    and three different rigs (mosaics of different sizes share a launch) - every mosaic and mask equal to the pair's own serial step;
    8 pairs = a chain of 6 and a chain of 2."""
    import torch
    from imagestitch_amd.pipeline import PairStitcher
    prec = {"i16": gpu.PREC_I16, "f32": gpu.PREC_F32, "f16acc32": gpu.PREC_F16ACC32}[prec_name]
    W, H, F, NP = 1280, 720, 1000.0, 8
    dev = torch.device("cuda:0")
    pairs, serial = [], []
    for p in range(NP):
        K, Rs = synth.camera_pair(W, H, F, yaw=(0.30, 0.36, 0.25)[p % 3])
        imgs = [torch.from_numpy(synth.make_tile(H, W, 1300 + 2 * p + i)).to(dev) for i in range(2)]
        ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, prec, 0, None, "int16")
        out, mask = ps.step()
        serial.append((out.clone(), mask.clone()))
        pairs.append(ps)
    assert len({tuple(s[0].shape) for s in serial}) == 3            # three mosaic sizes in the batch
    for rep in range(2):
        for ps in pairs:
            ps.out.fill_(-3); ps.out_mask.fill_(9)
        res = PairStitcher.step_batch(pairs)
        torch.cuda.synchronize()
        for p, ps in enumerate(pairs):
            ps.check_plan()
            assert torch.equal(res[p][1], serial[p][1]), (prec_name, rep, p)
            assert torch.equal(res[p][0], serial[p][0]), (prec_name, rep, p)


@pytest.mark.parametrize("branches,batch_size", [(1, 0), (3, 2)])
def test_batched_chain_captured_as_one_hipgraph(gpu, branches, batch_size):
    """BASELINE config 3 as ONE hipGraph: the batched launch chain of several pairs (PairStitcher.capture_batch) replayed - every mosaic equal
    to the pair's own serial step, the planned ROIs verified inside the graph (a stale plan raises the flag on replay).  branches = 3 (round 6):
    the same pairs as three parallel chains inside the one graph (side streams forked from and joined to the capture stream)."""
    import torch
    from imagestitch_amd._lib import IsxError
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F, NP = 1280, 720, 1000.0, 7
    dev = torch.device("cuda:0")
    pairs, serial = [], []
    for p in range(NP):
        K, Rs = synth.camera_pair(W, H, F, yaw=(0.30, 0.36)[p % 2])
        imgs = [torch.from_numpy(synth.make_tile(H, W, 1700 + 2 * p + i)).to(dev) for i in range(2)]
        ps = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16")
        out, mask = ps.step()
        serial.append((out.clone(), mask.clone()))
        pairs.append(ps)
    graph, gstream = PairStitcher.capture_batch(pairs, branches=branches, batch_size=batch_size)
    for rep in range(3):
        for ps in pairs:
            ps.out.fill_(-3); ps.out_mask.fill_(9)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for p, ps in enumerate(pairs):
            ps.check_plan()
            assert torch.equal(ps.out_mask, serial[p][1]) and torch.equal(ps.out, serial[p][0]), (rep, p)
    # new images, same rig: the replay reads the tensors in place
    pairs[3].imgs[0].copy_(torch.from_numpy(synth.make_tile(H, W, 4242)).to(dev))
    fresh = PairStitcher([t.clone() for t in pairs[3].imgs], pairs[3].K, pairs[3].Rs, F, "cylindrical", 5, gpu.PREC_F32, 0, None, "int16")
    exp = [t.clone() for t in fresh.step()]
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(pairs[3].out, exp[0]) and torch.equal(pairs[3].out_mask, exp[1])


def test_batched_blend_falls_back_for_blenders_that_do_not_qualify(gpu):
    """a batch that mixes a deferred multi-band blender, an eager one and a windowed one: all blended, each as alone"""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd.blender import blend_batch
    from imagestitch_amd.pipeline import PairStitcher
    W, H, F = 960, 540, 750.0
    dev = torch.device("cuda:0")
    K, Rs = synth.camera_pair(W, H, F)
    imgs = [torch.from_numpy(synth.make_tile(H, W, 1500 + i)).to(dev) for i in range(2)]
    ref = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, I.PREC_F32, 0, None, "int16")
    exp, exp_mask = [t.clone() for t in ref.step()]
    a = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, I.PREC_F32, 0, None, "int16")
    b = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, I.PREC_F32, 0, None, "int16", deferred=False)      # eager cycle
    c = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, I.PREC_I16, 0, None, "int16")                      # another precision
    for s in (a, b, c):
        s.step_until_blend()
    blend_batch([a.blender, b.blender, c.blender], [a.out, b.out, c.out], [a.out_mask, b.out_mask, c.out_mask])
    torch.cuda.synchronize()
    assert torch.equal(a.out, exp) and torch.equal(b.out, exp) and torch.equal(a.out_mask, exp_mask) and torch.equal(b.out_mask, exp_mask)
    ci = PairStitcher(imgs, K, Rs, F, "cylindrical", 5, I.PREC_I16, 0, None, "int16")
    assert torch.equal(c.out, ci.step()[0])
