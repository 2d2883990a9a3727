"""The N > 1 path on CPU: world_size 2 over gloo — pair sharding and the single all-gather that
assembles the mosaics (the same code path bench.py runs over RCCL)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagestitch_amd import mosaic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = mosaic.shard_pairs(n_pairs, world, rank)
    shapes = [(3 + p, 4, 3) for p in range(n_pairs)]
    cap = max(sum(int(np.prod(shapes[p])) for p in range(*mosaic.shard_pairs(n_pairs, world, r))) for r in range(world))
    # every "blended mosaic" of pair p is filled with the value p + 1
    mine = [torch.full(shapes[p], p + 1, dtype=torch.int16) for p in range(lo, hi)]
    send = mosaic.pack_blocks(mine, cap) if mine else torch.zeros((cap,), dtype=torch.int16)
    got = mosaic.gather_mosaics(send)
    ok = got.shape == (world, cap)
    for r in range(world):
        rlo, rhi = mosaic.shard_pairs(n_pairs, world, r)
        blocks = mosaic.unpack_blocks(got[r], [shapes[p] for p in range(rlo, rhi)])
        for p, b in zip(range(rlo, rhi), blocks):
            ok = ok and bool(torch.all(b == p + 1)) and tuple(b.shape) == shapes[p]
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_shard_pairs_partition():
    for n, w in ((32, 8), (5, 2), (3, 4), (0, 2), (7, 7)):
        spans = [mosaic.shard_pairs(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert mosaic.shard_pairs(32, 8, 3) == (12, 16)      # 64 tiles -> 4 pairs (8 tiles) per GPU


def test_all_gather_assembly_world2_gloo():
    world, n_pairs = 2, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_pairs, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def _worker_chunks(rank, world, port, n_pairs, ret):
    """bench.py's default schedule over gloo: the block travels pair by pair (mosaic.gather_chunk, one collective per pair into that pair's
    place of every rank's block), ragged shapes, 4-byte padded rows - world 8, 32 pairs = BASELINE config 4's sharding."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = mosaic.shard_pairs(n_pairs, world, rank)
    per = hi - lo
    ok = per == n_pairs // world
    # every rank's pair j is (5 + j) rows x (7 + j) px x 3 bytes, rows padded to 4 bytes, filled with rank * 16 + j + 1
    shapes = [(5 + j, 7 + j, 3) for j in range(per)]
    pitches = [(sh[1] * sh[2] + 3) // 4 * 4 for sh in shapes]
    n_out = sum(sh[0] * pt for sh, pt in zip(shapes, pitches))
    send = torch.zeros((n_out,), dtype=torch.uint8)
    gather_buf = torch.full((world * n_out,), 0x5a, dtype=torch.uint8)
    chunks, off = [], 0
    for j, (sh, pt) in enumerate(zip(shapes, pitches)):
        n = sh[0] * pt
        send[off:off + n].as_strided(sh, (pt, sh[2], 1)).fill_(rank * 16 + j + 1)
        chunks.append((off, n))
        off += n
    for (o, n) in chunks:
        mosaic.gather_chunk(send, o, n, gather_buf)
    for q in range(world):
        for j, ((o, n), sh, pt) in enumerate(zip(chunks, shapes, pitches)):
            got = mosaic.chunk_view(gather_buf, world, o, n, q).as_strided(sh, (pt, sh[2], 1))     # chunk-major: world * offset + q * count
            ok = ok and bool(torch.all(got == q * 16 + j + 1))
    # bench.py --gather root: the same chunks to rank 0 ONLY, in the same places; nobody else's buffer is touched
    gather_buf.fill_(0x5a)
    for (o, n) in chunks:
        mosaic.gather_chunk_root(send, o, n, gather_buf if rank == 0 else None, 0)
    if rank == 0:
        for q in range(world):
            for j, ((o, n), sh, pt) in enumerate(zip(chunks, shapes, pitches)):
                got = mosaic.chunk_view(gather_buf, world, o, n, q).as_strided(sh, (pt, sh[2], 1))
                ok = ok and bool(torch.all(got == q * 16 + j + 1))
    else:
        ok = ok and bool(torch.all(gather_buf == 0x5a))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_assembly_world8_gloo_32_pairs():
    """SURVEY §8(e) / BASELINE config 4 at its real world size: shard_pairs(32, 8, r) -> 4 pairs per rank, one whole-block all-gather and
    the pair-by-pair schedule, every rank's assembled batch checked on every rank (rank / offset bugs at N > 2 would otherwise meet the
    first 8-GPU run)."""
    world, n_pairs = 8, 32
    mgr = mp.Manager()
    for worker in (_worker, _worker_chunks):
        ret = mgr.dict()
        mp.spawn(worker, args=(world, _free_port(), n_pairs, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world)), (worker.__name__, dict(ret))
