"""Multi-GPU assembly of a batch of blended pairs (BASELINE config 4): one process per GPU, the
independent pairs are partitioned across ranks (no data-path collective while blending) and the
finished mosaics are assembled on every rank with ONE all-gather (RCCL over xGMI on the GPU box;
the same code runs over gloo on CPU tensors in the tests).
"""


def shard_pairs(n_pairs, world, rank):
    """Contiguous partition: rank r takes pairs [lo, hi).  64 tiles = 32 pairs on 8 GPUs -> 4 pairs each
    (SURVEY §8(e)); remainders go to the lowest ranks."""
    base, rem = divmod(n_pairs, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_blocks(tensors, capacity=None):
    """Flatten the rank's mosaics into one send block of `capacity` elements (zero padded so that every
    rank sends the same count, as all-gather requires)."""
    import torch
    n = sum(t.numel() for t in tensors)
    capacity = n if capacity is None else capacity
    assert capacity >= n
    out = torch.zeros((capacity,), dtype=tensors[0].dtype, device=tensors[0].device)
    off = 0
    for t in tensors:
        out[off:off + t.numel()].copy_(t.reshape(-1))
        off += t.numel()
    return out


def gather_mosaics(send, out=None, group=None):
    """ONE all-gather of every rank's packed block -> (world, capacity) on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * send.numel(),), dtype=send.dtype, device=send.device)
    # flat byte views: every backend (RCCL, gloo) takes uint8, whatever the pixel type is
    dist.all_gather_into_tensor(out.view(-1).view(torch.uint8), send.view(torch.uint8), group=group)
    return out.view(world, send.numel())


def unpack_blocks(gathered_row, shapes):
    """Inverse of pack_blocks for one rank's row."""
    out, off = [], 0
    for shp in shapes:
        n = 1
        for s in shp:
            n *= s
        out.append(gathered_row[off:off + n].reshape(shp))
        off += n
    return out


def write_mosaics(prefix, gathered, shapes_per_rank):
    """Tiled-mosaic writer for the gathered batch: one `<prefix>_r<rank>_p<pair>.bmp` per blended
    pair, from the (world, capacity) uint8 tensor gather_mosaics returns.  shapes_per_rank[r] = the (H, W, 3) shapes
    of rank r's mosaics in packing order.  Returns the file names (cv::imwrite W:315 per mosaic)."""
    from .imgio import imwrite
    names = []
    for r, shapes in enumerate(shapes_per_rank):
        for i, m in enumerate(unpack_blocks(gathered[r], shapes)):
            name = "%s_r%d_p%d.bmp" % (prefix, r, i)
            imwrite(name, m.contiguous() if hasattr(m, "contiguous") else m)
            names.append(name)
    return names
