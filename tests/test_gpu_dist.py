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
    # the direct schedule (isx_gather_p2p_*) at world 1: the rank's own buffer is its only destination
    ig = mosaic.IsxGather(0, collective=False)
    out = ig.p2p_setup(send.numel())
    out.zero_()
    ev = torch.cuda.Event(); ev.record()
    for off, n in chunks:
        ig.p2p_chunk(send, off, n, ev)
    ig.p2p_wait()
    torch.cuda.synchronize()
    for off, n in chunks:
        ok = ok and torch.equal(mosaic.chunk_view(out, world, off, n, rank), send[off:off + n])
    del out, ig
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


def _worker_p2p_world2(rank, world, port, ret):
    """Two processes on cuda:0 push their blocks into each other's receive buffers through HIP IPC (no RCCL: it refuses two ranks on
    one device; gloo carries the 64-byte handles and the barriers)."""
    import torch
    import torch.distributed as dist
    from imagestitch_amd import mosaic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(100 + rank)
    chunks = [(0, 4096), (4096, 70000), (74096, 1000)]
    n = sum(c[1] for c in chunks)
    send = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
    ig = mosaic.IsxGather(0, collective=False)
    buf = ig.p2p_setup(n)
    buf.zero_()
    torch.cuda.synchronize(); dist.barrier()              # nobody writes into a buffer that is still being cleared
    ev = torch.cuda.Event(); ev.record()
    for off, c in chunks:
        ig.p2p_chunk(send, off, c, ev)
    ig.p2p_synchronize()                                   # this rank's puts have landed ...
    dist.barrier()                                         # ... and so have everybody's
    torch.cuda.synchronize()
    ok = True
    for r in range(world):
        gr = torch.Generator(device=dev); gr.manual_seed(100 + r)
        expect = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=gr)
        for off, c in chunks:
            ok = ok and torch.equal(mosaic.chunk_view(buf, world, off, c, r), expect[off:off + c])
    ret[rank] = bool(ok)
    dist.barrier()
    del buf, ig
    dist.destroy_process_group()


def test_p2p_direct_gather_world2_on_one_gpu(gpu):
    import torch.multiprocessing as mp
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_p2p_world2, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def _worker_config4_rank_step(rank, world, port, ret):
    """BASELINE config 4's per-rank step at full size: 4 pairs of 4K tiles on 4 streams, each blend writing its CV_8UC3 mosaic straight
    into the pitched send block, the block gathered pair by pair behind each blend through RCCL (torch.distributed and the library's own
    communicator) and through the direct schedule; the gathered block equals the serial runs pair by pair and the oracle for pair 0."""
    import torch
    import torch.distributed as dist
    import imagestitch_amd as I
    from imagestitch_amd import mosaic, synth
    from imagestitch_amd.pipeline import PairStitcher
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    W4, H4, F4, NP = 3840, 2160, 3000.0, 4
    K, Rs = synth.camera_pair(W4, H4, F4)
    streams = [torch.cuda.Stream(device=dev) for _ in range(NP)]
    host_imgs = [[synth.make_tile(H4, W4, 40 * p + i) for i in range(2)] for p in range(NP)]
    serial = []
    for p in range(NP):                                     # the reference: every pair alone, on the default stream
        ps = PairStitcher([torch.from_numpy(i).to(dev) for i in host_imgs[p]], K, Rs, F4, "cylindrical", 5, I.PREC_F32, 0, None, "uint8")
        out, _ = ps.step()
        serial.append(out.cpu().numpy().copy())
        del ps
    pairs = [PairStitcher([torch.from_numpy(i).to(dev) for i in host_imgs[p]], K, Rs, F4, "cylindrical", 5, I.PREC_F32, 0, streams[p], "uint8")
             for p in range(NP)]
    shapes = [tuple(p.out.shape) for p in pairs]
    pitches = [(sh[1] * sh[2] + 3) // 4 * 4 for sh in shapes]
    sizes = [sh[0] * pt for sh, pt in zip(shapes, pitches)]
    n_out = sum(sizes)
    send = torch.empty((n_out,), dtype=torch.uint8, device=dev)
    chunks, off = [], 0
    for p, (sh, pt, n) in enumerate(zip(shapes, pitches, sizes)):
        pairs[p].out = send[off:off + n].as_strided(sh, (pt, sh[2], 1))
        chunks.append((off, n)); off += n
    ok = True
    for backend in ("torch", "isx", "p2p"):
        ig = None
        if backend == "isx":
            ig = mosaic.IsxGather(0)
            gbuf = torch.zeros((world * n_out,), dtype=torch.uint8, device=dev)
        elif backend == "p2p":
            ig = mosaic.IsxGather(0, collective=False)
            gbuf = ig.p2p_setup(n_out); gbuf.zero_()
        else:
            gbuf = torch.zeros((world * n_out,), dtype=torch.uint8, device=dev)
        send.fill_(7)
        torch.cuda.synchronize()
        comm = torch.cuda.Stream(device=dev)
        evs = [torch.cuda.Event() for _ in range(NP)]
        for step in range(2):                               # the second step replays the planned one
            for p in range(NP):
                with torch.cuda.stream(streams[p]):
                    pairs[p].step()
                evs[p].record(streams[p])
                o, n = chunks[p]
                if backend == "p2p":
                    ig.p2p_chunk(send, o, n, evs[p])
                elif backend == "isx":
                    ig.chunk(send, o, n, gbuf, evs[p])
                else:
                    comm.wait_event(evs[p])
                    with torch.cuda.stream(comm):
                        mosaic.gather_chunk(send, o, n, gbuf)
            if backend == "p2p":
                ig.p2p_wait(comm)
            elif backend == "isx":
                ig.wait(comm)
            comm.synchronize()
            for p in range(NP):
                streams[p].wait_stream(comm)              # the next step overwrites the send block
        torch.cuda.synchronize()
        for p in range(NP):
            pairs[p].check_plan()
            o, n = chunks[p]
            got = mosaic.chunk_view(gbuf, world, o, n, rank).as_strided(shapes[p], (pitches[p], shapes[p][2], 1)).cpu().numpy()
            ok = ok and np.array_equal(got, serial[p])
        del gbuf, ig
    # pair 0 against the oracle (fp32 blend converted to CV_8U: blend + convertTo, W:315)
    from oracle import capi as O
    corners, warped, wmasks = [], [], []
    for im, R in zip(host_imgs[0], Rs):
        c, wi, _ = O.warp_u8(O.CYL, F4, K, R, im, O.LINEAR, O.BORDER_REFLECT)
        _, wm, _ = O.warp_u8(O.CYL, F4, K, R, np.full((H4, W4), 255, np.uint8), O.NEAREST, O.BORDER_CONSTANT)
        corners.append(c); warped.append(wi); wmasks.append(wm)
    seam = synth.seam_masks(corners, wmasks)
    mb = O.MultiBand(5, O.F32)
    mb.prepare(corners, [(m.shape[1], m.shape[0]) for m in wmasks])
    for wi, sm, c in zip(warped, seam, corners):
        mb.feed(wi.astype(np.int16), sm, c)
    of32, om = mb.blend(True)
    o8 = np.clip(np.rint(of32), 0, 255).astype(np.uint8)     # saturate_cast<uchar>(cvRound(v)): round half to even
    ok = ok and np.array_equal(serial[0], o8)
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_config4_per_rank_step_at_4k_with_chunked_gathers(gpu):
    import torch.multiprocessing as mp
    ret = mp.Manager().dict()
    mp.spawn(_worker_config4_rank_step, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret.get(0), dict(ret)


@pytest.mark.parametrize("backend,gather,shard", [("p2p", "chunk", "pairs"), ("p2p", "single", "pairs"), ("torch", "chunk", "pairs"), ("torch", "single", "pairs"),
                                                  ("p2p", "single", "strips")])
def test_bench_walks_its_n2_path_on_one_gpu(gpu, tmp_path, backend, gather, shard):
    """bench.py's whole N > 1 branch (rank / world indexing, the send blocks, chunk offsets and events, the three timed legs, the reduction of
    the timings, rank 0's JSON line) at world size 2 on ONE GPU: both ranks on cuda:0 (ISX_BENCH_ONE_GPU=1: gloo for barriers and
    reductions), the gathers real device copies through HIP IPC (--gather-backend p2p) or torch.distributed's all-gather over gloo (the
    default backend's code path with another transport), pair by pair and as one collective per step.  What a one-GPU box can check of the
    command the driver runs on 2 / 4 / 8 GPUs; RCCL itself is covered at world 1 (test_config4_per_rank_step_at_4k_with_chunked_gathers)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--pairs", "2", "--width", "1280", "--height", "720", "--focal", "1000",
           "--gather-backend", backend, "--gather", gather, "--no-cpu-baseline", "--no-dropin", "--no-live-traffic"]
    if shard == "strips":      # ONE panorama of 6 tiles per step, cut into two column strips (strong scaling)
        cmd += ["--shard", "strips", "--tiles", "6", "--yaw", "0.275"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == ("weak" if shard == "pairs" else "strong") and d["value"] > 0 and d["config"]["pairs_per_gpu"] == 2
    mg = d["multi_gpu"]
    assert mg["gather_backend"] == backend and mg["gather"] == gather and mg["send_bytes_per_rank"] > 0 and mg["gather_bus_GBs"] > 0 and mg["without_gather_Mpix_s"] > 0



@pytest.mark.parametrize("backend,gather,shard", [("p2p", "chunk", "pairs"), ("torch", "single", "pairs"), ("torch", "chunk", "pairs"), ("p2p", "single", "pairs"),
                                                  ("p2p", "chunk", "strips"), ("torch", "single", "strips"), ("torch", "root", "pairs")])
def test_bench_walks_its_n8_path_on_one_gpu(gpu, tmp_path, backend, gather, shard):
    """The rehearsal of the first 8-GPU run: bench.py at WORLD SIZE 8 on one GPU (ISX_BENCH_ONE_GPU=1), BASELINE config 4's
    real shape - 32 pairs, 4 per rank (small tiles) - through the direct schedule (HIP IPC between 8 processes: every rank maps 7 peers'
    buffers and pushes to them in its neighbour order) and through torch.distributed's collectives (gloo transport), pair by pair and as one
    collective; and ONE panorama of 64 tiles cut into 8 column strips.  --check-gather: every rank compares all 32 chunks (8 strips) it
    received with that mosaic stitched serially from the owning rank's seed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ISX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "2", "--preflight-ms", "0",
           "--gather-backend", backend, "--gather", gather, "--check-gather", "--no-cpu-baseline", "--no-dropin", "--no-live-traffic"]
    if shard == "strips":      # ONE panorama of 64 tiles (config 4's tile count) per step, cut into eight column strips (strong scaling)
        cmd += ["--shard", "strips", "--tiles", "64", "--pairs", "1", "--width", "320", "--height", "180", "--focal", "800", "--yaw", "0.045", "--bands", "3"]
    else:
        cmd += ["--pairs", "4", "--width", "640", "--height", "360", "--focal", "500"]
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == ("weak" if shard == "pairs" else "strong") and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["rccl_ranks"] == 8 and mg["gather_backend"] == backend and mg["gather"] == gather
    chk = mg["gather_check"]
    assert chk["chunks_per_rank"] == (32 if shard == "pairs" else 8) and chk["mismatched_over_all_ranks"] == 0, chk
    # the second reported leg: every chunk to rank 0 only - present whenever the graded schedule is an all-gather over torch.distributed
    assert ("root_gather_Mpix_s" in mg) == (backend == "torch" and gather != "root"), mg
    if shard == "pairs":
        assert d["config"]["pairs_per_gpu"] == 4 and d["config"]["tiles_per_mosaic"] == 2
        # the send block: 4 mosaics of CV_8UC3 rows padded to 4 bytes
        px = d["config"]["mosaic_px"]
        assert mg["send_bytes_per_rank"] % 4 == 0 and 4 * 3 * px * 0.7 <= mg["send_bytes_per_rank"] <= 4 * 3 * px      # (the result is the padded mosaic - mosaic_px - cropped to the tiles' union)
    else:
        assert d["config"]["tiles_per_mosaic"] == 64 and d["config"]["strip"] == "0/8"
