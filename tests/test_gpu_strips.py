"""One panorama of many tiles cut into column strips ("shard by contiguous strips and have each GPU also feed its
neighbour tiles (halo recompute) so that no exchange is needed before the final gather").

* MultiBandBlender.set_window: the columns [x0, x1) of a deferred blend, computed from only the tiles that can reach them, equal the
  same columns of the whole blend bit for bit - every precision, 3 and 5 bands, strips for 2 / 3 / 5 ranks;
* the whole blend those strips are compared with equals the oracle's;
* world 2 (both ranks on cuda:0, gloo for the gather of host copies): every rank warps and blends its strip of a ring of 6 tiles, the
  all-gather of the strips IS the panorama on every rank;
* the error contract of isx_blender_set_window.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, F, N_TILES, YAW_STEP = 480, 270, 380.0, 6, 0.62


def _tiles():
    from imagestitch_amd import synth
    K, Rs = synth.camera_ring(W, H, F, N_TILES, YAW_STEP)
    return K, Rs, [synth.make_tile(H, W, 300 + i) for i in range(N_TILES)]


def _stitcher(torch, imgs, K, Rs, bands, prec, window=None, out_dtype="int16", only=None):
    from imagestitch_amd.pipeline import MosaicStitcher
    dev = torch.device("cuda:0")
    dimgs = [torch.from_numpy(im).to(dev) if (only is None or i in only) else None for i, im in enumerate(imgs)]
    return MosaicStitcher(dimgs, K, Rs, F, "cylindrical", bands, prec, 0, None, out_dtype, window=window)


@pytest.mark.parametrize("prec_name", ["i16", "f32", "f16acc32"])
@pytest.mark.parametrize("bands", [3, 5])
def test_strips_equal_the_whole_blend(gpu, prec_name, bands):
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import mosaic
    prec = {"i16": I.PREC_I16, "f32": I.PREC_F32, "f16acc32": I.PREC_F16ACC32}[prec_name]
    K, Rs, imgs = _tiles()
    whole = _stitcher(torch, imgs, K, Rs, bands, prec)
    full, full_mask = [t.cpu().numpy() for t in whole.step()]
    fw, fh = whole.mosaic_size
    assert full.shape == (fh, fw, 3) and fw > 5 * 128
    fewer = 0
    for world in (2, 3, 5):
        windows, sw = mosaic.strip_windows(fw, world)
        assert sw % 128 == 0 and windows[-1][1] >= fw
        for x0, x1 in windows:
            if x1 == x0:
                continue
            st = _stitcher(torch, imgs, K, Rs, bands, prec, window=(x0, x1))
            fewer += len(st.active) < N_TILES
            st.out.fill_(-7); st.out_mask.fill_(7)            # columns past the mosaic's edge must stay as they are
            for i in st.active:                                # a step re-warps only the columns the strip depends on: whatever
                st.warped[i].fill_(171)                        # else the tiles hold must not matter
                c0, c1 = st.tile_cols[i]
                assert 0 <= c0 < c1 <= st.sizes[i][0]
            out, mask = [t.cpu().numpy() for t in st.step()]
            out2 = st.step()[0].cpu().numpy()                  # the planned step replayed: same strip
            st.check_plan()
            xe = min(x1, fw)
            assert out.shape == (fh, sw, 3)
            assert np.array_equal(out[:, :xe - x0], full[:, x0:xe]), (world, x0, x1, st.active)
            assert np.array_equal(mask[:, :xe - x0], full_mask[:, x0:xe])
            assert np.array_equal(out2, out)
            assert (out[:, xe - x0:] == -7).all() and (mask[:, xe - x0:] == 7).all()
    assert fewer >= 6      # most strips are computed from a subset of the tiles


def test_the_whole_blend_of_the_ring_equals_the_oracle(gpu):
    """anchors the comparison above: the unwindowed HIP mosaic of the 6-tile ring is the oracle's"""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import synth
    from oracle import capi as O
    K, Rs, imgs = _tiles()
    whole = _stitcher(torch, imgs, K, Rs, 5, I.PREC_I16)
    full, full_mask = [t.cpu().numpy() for t in whole.step()]
    corners, warped, wmasks = [], [], []
    for im, R in zip(imgs, Rs):
        c, wi, _ = O.warp_u8(O.CYL, F, K, R, im, O.LINEAR, O.BORDER_REFLECT)
        _, wm, _ = O.warp_u8(O.CYL, F, K, R, np.full((H, W), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    seam = synth.seam_masks(corners, wmasks)
    mb = O.MultiBand(5, O.I16)
    mb.prepare(corners, [(m.shape[1], m.shape[0]) for m in wmasks])
    for wi, sm, c in zip(warped, seam, corners):
        mb.feed(wi.astype(np.int16), sm, c)
    ref, ref_mask = mb.blend(False)
    assert np.array_equal(full, ref) and np.array_equal(full_mask, ref_mask)


def test_twelve_tile_ring_whole_and_strips(gpu):
    """more tiles than round 1's deferred cycle held (8): the whole 12-tile mosaic equals the oracle's, every strip equals the whole"""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import mosaic, synth
    from imagestitch_amd.pipeline import MosaicStitcher
    from oracle import capi as O
    n, w, h, f, step = 12, 320, 180, 300.0, 0.45
    K, Rs = synth.camera_ring(w, h, f, n, step)
    imgs = [synth.make_tile(h, w, 500 + i) for i in range(n)]
    dev = torch.device("cuda:0")
    dimgs = [torch.from_numpy(im).to(dev) for im in imgs]
    whole = MosaicStitcher(dimgs, K, Rs, f, "cylindrical", 4, I.PREC_I16, 0, None, "int16")
    full, full_mask = [t.cpu().numpy() for t in whole.step()]
    corners, warped, wmasks = [], [], []
    for im, R in zip(imgs, Rs):
        c, wi, _ = O.warp_u8(O.CYL, f, K, R, im, O.LINEAR, O.BORDER_REFLECT)
        _, wm, _ = O.warp_u8(O.CYL, f, K, R, np.full((h, w), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    seam = synth.seam_masks(corners, wmasks)
    mb = O.MultiBand(4, O.I16)
    mb.prepare(corners, [(m.shape[1], m.shape[0]) for m in wmasks])
    for wi, sm, c in zip(warped, seam, corners):
        mb.feed(wi.astype(np.int16), sm, c)
    ref, ref_mask = mb.blend(False)
    assert np.array_equal(full, ref) and np.array_equal(full_mask, ref_mask)
    fw, fh = whole.mosaic_size
    windows, sw = mosaic.strip_windows(fw, 4)
    for x0, x1 in windows:
        st = MosaicStitcher(dimgs, K, Rs, f, "cylindrical", 4, I.PREC_I16, 0, None, "int16", window=(x0, x1))
        assert len(st.active) < n
        out = st.step()[0].cpu().numpy()
        xe = min(x1, fw)
        assert np.array_equal(out[:, :xe - x0], full[:, x0:xe])


def test_a_strip_needs_only_its_neighbourhood(gpu):
    """tiles_for_window is tight enough to matter and safe: dropping a tile it lists changes the strip, and it never lists all 6."""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import mosaic
    K, Rs, imgs = _tiles()
    whole = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32)
    fw, _ = whole.mosaic_size
    windows, _ = mosaic.strip_windows(fw, 6)
    counts = []
    assert windows[-1] == (windows[-1][0], windows[-1][0])      # 6 strips of 384 columns: the last one starts past the mosaic
    for x0, x1 in windows:
        if x1 == x0:
            continue
        act = mosaic.tiles_for_window(whole.corners, whole.sizes, 5, x0, x1)
        counts.append(len(act))
        # only the listed tiles resident on this "rank"
        st = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, window=(x0, x1), only=set(act))
        assert st.active == act
    assert max(counts) < N_TILES and min(counts) >= 1 and sum(counts) < 3 * N_TILES, counts      # here: [3, 4, 5, 3, 1] (tiles 480 wide, gap 96)
    with pytest.raises(ValueError):
        x0, x1 = windows[2]
        act = mosaic.tiles_for_window(whole.corners, whole.sizes, 5, x0, x1)
        _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, window=(x0, x1), only=set(act[1:]))


def test_window_u8_strip_into_a_pitched_send_block(gpu):
    """what a rank actually sends: the strip as CV_8UC3 (blend + convertTo(CV_8U)) written into a slice of a packed block"""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import mosaic
    K, Rs, imgs = _tiles()
    whole = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, out_dtype="uint8")
    full = whole.step()[0].cpu().numpy()
    fw, fh = whole.mosaic_size
    windows, sw = mosaic.strip_windows(fw, 3)
    strips = []
    for x0, x1 in windows:
        st = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, window=(x0, x1), out_dtype="uint8")
        st.out = torch.zeros((fh, sw, 3), dtype=torch.uint8, device="cuda:0")     # a contiguous strip of the send block
        strips.append(st.step()[0].reshape(-1).clone())
    pano = mosaic.assemble_strips(torch.stack(strips), fh, sw, fw).cpu().numpy()
    assert np.array_equal(pano, full)


def test_window_error_contract(gpu):
    import torch
    import imagestitch_amd as I
    from imagestitch_amd._lib import IsxError
    from imagestitch_amd.blender import MultiBandBlender
    b = MultiBandBlender(False, 3, I.PREC_F32, 0)
    with pytest.raises(IsxError):
        b.set_window(100, 400)            # first column not a multiple of ISX_WINDOW_GRANULE
    with pytest.raises(IsxError):
        b.set_window(256, 256)
    b.set_window(0, 0)                    # no window
    dev = torch.device("cuda:0")
    img = torch.full((64, 300, 3), 90, dtype=torch.uint8, device=dev)
    msk = torch.full((64, 300), 255, dtype=torch.uint8, device=dev)
    # eager cycle: a window is refused at blend()
    b.set_window(128, 256)
    b.prepare([(0, 0)], [(300, 64)])
    b.feed_u8(img, msk, (0, 0))
    with pytest.raises(IsxError):
        b.blend()
    # deferred cycle: mats of the window's width are required
    b.prepare([(0, 0)], [(300, 64)])      # a new cycle (the refused blend() left the old one as it was)
    b.set_deferred_level0(True)
    b.feed_u8(img, msk, (0, 0))
    with pytest.raises(IsxError):
        b.blend(torch.empty((64, 300, 3), dtype=torch.int16, device=dev), torch.empty((64, 300), dtype=torch.uint8, device=dev))
    out, mask = b.blend()
    assert tuple(out.shape) == (64, 128, 3) and (out.cpu().numpy() == 90).all() and (mask.cpu().numpy() == 255).all()
    # a window that starts past the result
    b.set_window(384, 512)
    b.prepare([(0, 0)], [(300, 64)])
    b.feed_u8(img, msk, (0, 0))
    with pytest.raises(IsxError):
        b.blend()


def test_window_and_column_range_with_host_mats_copy_back_only_the_computed_columns(gpu):
    """host (numpy) outputs: the staged device copy holds only the computed columns, so only those are copied back - the padded
    columns of a last strip and the columns outside isx_warper_set_dst_columns keep what the caller's mat held"""
    import torch
    import imagestitch_amd as I
    from imagestitch_amd import synth
    from imagestitch_amd.blender import MultiBandBlender
    rng = np.random.default_rng(77)
    img = rng.integers(0, 255, (80, 300, 3), dtype=np.uint8)
    msk = np.full((80, 300), 255, np.uint8)

    def cycle(window, out, om):
        b = MultiBandBlender(False, 3, I.PREC_F32, 0)
        if window:
            b.set_window(*window)
        b.prepare([(0, 0)], [(300, 80)])
        b.set_deferred_level0(True)
        b.feed_u8(img, msk, (0, 0))
        return b.blend(out, om)

    full, fmask = cycle(None, None, None)
    assert isinstance(full, np.ndarray) and full.shape == (80, 300, 3)
    out, om = np.full((80, 128, 3), -7, np.int16), np.full((80, 128), 9, np.uint8)
    cycle((256, 384), out, om)            # the result is 300 wide: 44 valid columns, 84 of padding
    assert np.array_equal(out[:, :44], full[:, 256:300]) and np.array_equal(om[:, :44], fmask[:, 256:300])
    assert (out[:, 44:] == -7).all() and (om[:, 44:] == 9).all()

    K, Rs = synth.camera_pair(500, 300, 380.0, yaw=0.3)
    src = synth.make_tile(300, 500, 3, noise_only=True)
    wp = I.CylindricalWarper().create(380.0)
    corner, wfull, wmask = wp.warp_with_mask(src, K, Rs[0])
    assert isinstance(wfull, np.ndarray)
    h, w = wfull.shape[:2]
    for c0, c1 in ((70, 200), (0, 1), (w - 5, w), (130, w + 40)):
        di, dm = np.full((h, w, 3), 77, np.uint8), np.full((h, w), 9, np.uint8)
        wp.set_dst_columns(c0, c1)
        assert wp.warp_with_mask(src, K, Rs[0], dst_img=di, dst_mask=dm)[0] == corner
        wp.set_dst_columns(0, 0)
        lo, hi = c0 // 64 * 64, min(c1, w)
        assert np.array_equal(di[:, lo:hi], wfull[:, lo:hi]) and np.array_equal(dm[:, lo:hi], wmask[:, lo:hi]), (c0, c1)
        assert (di[:, :lo] == 77).all() and (dm[:, :lo] == 9).all() and (di[:, hi:] == 77).all() and (dm[:, hi:] == 9).all(), (c0, c1)


def test_feather_and_multiband_strip_fuzz_slice(gpu):
    """a few hundred random strips of random tile rows against the ORACLE's whole blends (tools/fuzz_parity.py: case_strip,
    case_strip_feather), every precision and input type"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity as F
    F.G.load()
    ran = 0
    for n in range(200):
        ran += F.case_strip(np.random.default_rng(4242000 + n)) != "skip"
        ran += F.case_strip_feather(np.random.default_rng(4343000 + n)) != "skip"
    assert ran > 300


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker_strips(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    import imagestitch_amd as I
    from imagestitch_amd import mosaic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    K, Rs, imgs = _tiles()
    # geometry first (every rank derives the same strips from the rig alone), then only this rank's tiles go to the device
    plan = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, out_dtype="uint8")
    fw, fh = plan.mosaic_size
    windows, sw = mosaic.strip_windows(fw, world)
    x0, x1 = windows[rank]
    act = mosaic.tiles_for_window(plan.corners, plan.sizes, 5, x0, x1)
    st = _stitcher(torch, imgs, K, Rs, 5, I.PREC_F32, window=(x0, x1), out_dtype="uint8", only=set(act))
    st.out = torch.zeros((fh, sw, 3), dtype=torch.uint8, device="cuda:0")
    st.step()
    strip = st.step()[0]
    st.check_plan()
    got = mosaic.gather_mosaics(strip.reshape(-1).cpu())            # (world, fh * sw * 3) on every rank
    pano = mosaic.assemble_strips(got, fh, sw, fw).numpy()
    full = plan.step()[0].cpu().numpy()                              # the unsharded mosaic, for the check only
    ret[rank] = bool(np.array_equal(pano, full)) and len(act) < N_TILES
    dist.barrier()
    dist.destroy_process_group()


def test_world2_strips_of_one_panorama_assemble_it(gpu):
    import torch.multiprocessing as mp
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_strips, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_example_stitch_strips_three_ranks(gpu, tmp_path):
    """examples/stitch_strips.py under torch.distributed.run: 3 ranks share the one GPU (gloo), rank 0 checks the assembled panorama
    against the whole blend and writes it"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "pano.bmp")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "examples", "stitch_strips.py"), "--backend", "gloo", "--check",
                        "--tiles", "6", "--width", "960", "--height", "540", "--focal", "750", "--out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "strips == whole blend: True" in r.stdout
    import imagestitch_amd as I
    pano = I.imread(out)
    assert pano.shape[2] == 3 and pano.shape[1] > 3 * 128
