"""The N > 1 path with the REAL blend (BASELINE config 4) as far as a one-GPU box allows:

* world 2, both ranks on cuda:0: every rank stitches its shard of a batch of pairs with the HIP library, the blended mosaics are
  assembled on every rank by all-gather (gloo on host copies - RCCL refuses two ranks on one device), and every rank's assembled batch
  equals the oracle's mosaics, pair by pair;
* world 1 over RCCL: the library's own communicator (isx_gather_*: the C-ABI form of the collective) and torch.distributed's, whole
  block and chunk by chunk, deliver the bytes that were sent.
"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, F, BANDS, N_PAIRS = 640, 360, 500.0, 5, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pair_inputs(p):
    from imagestitch_amd import synth
    K, Rs = synth.camera_pair(W, H, F, yaw=0.30 + 0.01 * p)
    return K, Rs, [synth.make_tile(H, W, 900 + 2 * p + i) for i in range(2)]


def _oracle_mosaic(p):
    from imagestitch_amd import synth
    from oracle import capi as O
    K, Rs, imgs = _pair_inputs(p)
    corners, warped, wmasks = [], [], []
    for im, R in zip(imgs, Rs):
        c, wi, _ = O.warp_u8(O.CYL, F, K, R, im, O.LINEAR, O.BORDER_REFLECT)
        _, wm, _ = O.warp_u8(O.CYL, F, K, R, np.full((H, W), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    seam = synth.seam_masks(corners, wmasks)
    mb = O.MultiBand(BANDS, O.I16)
    mb.prepare(corners, [(m.shape[1], m.shape[0]) for m in wmasks])
    for wi, sm, c in zip(warped, seam, corners):
        mb.feed(wi.astype(np.int16), sm, c)
    return mb.blend(False)[0]


def _worker_world2(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    import imagestitch_amd as I
    from imagestitch_amd import mosaic
    from imagestitch_amd.pipeline import PairStitcher
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    lo, hi = mosaic.shard_pairs(N_PAIRS, world, rank)
    mine = []
    for p in range(lo, hi):                                    # this rank's pairs through the HIP library (planned step + a replayed one)
        K, Rs, imgs = _pair_inputs(p)
        ps = PairStitcher([torch.from_numpy(i).to(dev) for i in imgs], K, Rs, F, "cylindrical", BANDS, I.PREC_I16, 0, None, "int16")
        ps.step()
        out, _ = ps.step()
        ps.check_plan()
        mine.append(out.cpu().contiguous())
    expect = [_oracle_mosaic(p) for p in range(N_PAIRS)]      # every rank checks the WHOLE assembled batch
    shapes = [tuple(e.shape) for e in expect]
    cap = max(sum(int(np.prod(shapes[p])) for p in range(*mosaic.shard_pairs(N_PAIRS, world, r))) for r in range(world))
    send = mosaic.pack_blocks(mine, cap)
    got = mosaic.gather_mosaics(send)                           # ONE all-gather: (world, cap) on every rank
    ok = True
    for r in range(world):
        rlo, rhi = mosaic.shard_pairs(N_PAIRS, world, r)
        for p, blk in zip(range(rlo, rhi), mosaic.unpack_blocks(got[r], shapes[rlo:rhi])):
            ok = ok and np.array_equal(blk.numpy(), expect[p])
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_world2_sharded_real_blend_assembles_the_oracles_batch(gpu):
    import torch.multiprocessing as mp
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_world2, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _worker_rccl_world1(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    from imagestitch_amd import mosaic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g = torch.Generator(device=dev); g.manual_seed(9)
    send = torch.randint(0, 256, (3 * 1000 + 8,), dtype=torch.uint8, device=dev, generator=g)
    chunks = [(0, 1000), (1000, 1000), (2000, 1008)]
    ok = True
    for backend in ("torch", "isx"):
        ig = mosaic.IsxGather(0) if backend == "isx" else None
        out = torch.zeros((world * send.numel(),), dtype=torch.uint8, device=dev)
        if ig is not None:
            ig.all(send, out)
        else:
            mosaic.gather_mosaics(send, out)
        torch.cuda.synchronize()
        ok = ok and torch.equal(out.view(world, -1)[rank], send)
        out.zero_()
        ev = torch.cuda.Event(); ev.record()
        comm = torch.cuda.Stream(device=dev)
        for off, n in chunks:
            if ig is not None:
                ig.chunk(send, off, n, out, ev)
            else:
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    mosaic.gather_chunk(send, off, n, out)
        if ig is not None:
            ig.wait()
            ig.synchronize()
        torch.cuda.synchronize()
        for off, n in chunks:
            ok = ok and torch.equal(mosaic.chunk_view(out, world, off, n, rank), send[off:off + n])
        del ig
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_gather_whole_block_and_chunks_world1(gpu):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_worker_rccl_world1, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret.get(0), dict(ret)
