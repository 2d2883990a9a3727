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
