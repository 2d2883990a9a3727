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


def gather_chunk(send, offset, count, out, group=None):
    """The all-gather of ONE chunk (one pair's mosaic) of the rank's send block: elements [offset, offset + count) of every
    rank's block land rank-major at out[world * offset : world * (offset + count)] - the layout isx_gather_chunk uses.  Posting
    the chunks one by one as their blends are enqueued (each on a communication stream behind that blend's event) hides all but
    the last pair's transfer under compute; every rank must post the same chunks in the same order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dst = out.view(-1).view(torch.uint8)[world * offset: world * (offset + count)]
    dist.all_gather_into_tensor(dst, send.view(-1).view(torch.uint8)[offset: offset + count], group=group)
    return dst.view(world, count)


def chunk_view(out, world, offset, count, rank):
    """Rank `rank`'s copy of the chunk at (offset, count) inside a buffer filled by gather_chunk / isx_gather_chunk."""
    return out.view(-1)[world * offset + rank * count: world * offset + (rank + 1) * count]


class IsxGather:
    """The same collective through the C-ABI (isx_gather_*: an RCCL communicator owned by the library, for pipelines without
    torch).  The 128-byte rendezvous id travels over whatever the caller has; here: torch.distributed's object broadcast."""

    def __init__(self, device, group=None):
        import ctypes as C
        import torch.distributed as dist
        from . import _lib
        self._lib, self._C = _lib.load(), C
        self._check = _lib.check
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            self._check(self._lib.isx_gather_unique_id(buf))
            ident[0] = buf.raw
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        self._h = C.c_void_p()
        self._check(self._lib.isx_gather_create(self.world, self.rank, ident[0], int(device), C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.isx_gather_destroy(h)
            self._h = None

    def all(self, send, out, stream=None):
        """ONE all-gather of the whole block on `stream` (default: the current torch stream)."""
        import torch
        st = stream if stream is not None else torch.cuda.current_stream()
        self._check(self._lib.isx_gather_all(self._h, self._C.c_void_p(send.data_ptr()), send.numel() * send.element_size(),
                                             self._C.c_void_p(out.data_ptr()), self._C.c_void_p(st.cuda_stream)))
        return out.view(self.world, -1)

    def chunk(self, send, offset, count, out, ready_event=None):
        """isx_gather_chunk: bytes [offset, offset + count) of the block, on the handle's communication stream behind ready_event."""
        ev = self._C.c_void_p(ready_event.cuda_event) if ready_event is not None else None
        self._check(self._lib.isx_gather_chunk(self._h, self._C.c_void_p(send.data_ptr()), send.numel() * send.element_size(), offset, count,
                                               self._C.c_void_p(out.data_ptr()), ev))

    def wait(self, stream=None):
        import torch
        st = stream if stream is not None else torch.cuda.current_stream()
        self._check(self._lib.isx_gather_wait(self._h, self._C.c_void_p(st.cuda_stream)))

    def synchronize(self):
        self._check(self._lib.isx_gather_synchronize(self._h))


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
