"""The hot path as the reference's main() drives it (W:223-233, 281, 294-302, 313), for the tiles of one
mosaic on one GPU (a pair in BASELINE configs 1-4, a row of 8 in config 5): warp(image) + warp(mask) per
tile -> prepare -> feed x n -> blend.

PairStitcher owns the device buffers so that the steady state allocates nothing.  Seam masks (the
output of the seam finder, which is out of scope: SURVEY §8(f) N1) are inputs: they are built
once from the warped masks with synth.seam_masks and stay resident.
"""
import os

import numpy as np

from . import _lib, synth
from .blender import MultiBandBlender
from .warper import CylindricalWarper, SphericalWarper


def feed_geometry(roi, num_bands, tl, size):
    """Padded tile rectangle of MultiBandBlender::feed (A11) -> (width, height) of the tile pyramid base."""
    rx, ry, rw, rh = roi
    L, m = num_bands, 1 << num_bands
    gap = 3 * m
    tlx, tly = max(rx, tl[0] - gap), max(ry, tl[1] - gap)
    brx, bry = min(rx + rw, tl[0] + size[0] + gap), min(ry + rh, tl[1] + size[1] + gap)
    tlx = rx + (((tlx - rx) >> L) << L)
    tly = ry + (((tly - ry) >> L) << L)
    w, h = brx - tlx, bry - tly
    w += (m - w % m) % m
    h += (m - h % m) % m
    return w, h


def prepare_geometry(corners, sizes, num_bands):
    """dst_roi_ of MultiBandBlender::prepare (A9) -> ((x, y, padded_w, padded_h), (final_w, final_h), L)."""
    c = np.asarray(corners).reshape(-1, 2)
    s = np.asarray(sizes).reshape(-1, 2)
    tl, br = c.min(0), (c + s).max(0)
    w, h = int(br[0] - tl[0]), int(br[1] - tl[1])
    L = min(num_bands, int(np.ceil(np.log(float(max(w, h))) / np.log(2.0))))
    m = 1 << L
    return (int(tl[0]), int(tl[1]), w + (m - w % m) % m, h + (m - h % m) % m), (w, h), L


def model_bytes(src_px, warped_px, tile_base_px, mosaic_px, precision, num_bands):
    """ALGORITHMIC bytes of one pair: each pyramid level materialised once,
    elementwise work fused into its stencil, level-0 inputs at native width (u8x3 image + u8 mask
    into feed — the convertTo(CV_16S) of W:294 is fused), maps never materialised.
    Returns a dict with warp / feed / blend / total bytes."""
    g = {_lib.PREC_F32: 16.0, _lib.PREC_I16: 10.0, _lib.PREC_F16ACC32: 8.0}[precision]
    g_rgb = 12.0 if precision == _lib.PREC_F32 else 6.0
    d = 10.0 if precision == _lib.PREC_I16 else 16.0
    d_rgb = 6.0 if precision == _lib.PREC_I16 else 12.0
    L = num_bands
    warp = sum(3.0 * s + 4.0 * n for s, n in zip(src_px, warped_px))
    feed = 0.0
    for p in tile_base_px:
        lv = [p / 4.0 ** k for k in range(L + 1)]
        for k in range(L):                      # pyrDown k -> k+1
            feed += lv[k] * (4.0 if k == 0 else g) + lv[k + 1] * g
        for k in range(L):                      # Laplacian + accumulate level k
            feed += lv[k] * ((4.0 if k == 0 else g) + 2.0 * d) + lv[k + 1] * g_rgb
        feed += lv[L] * ((4.0 if L == 0 else g) + 2.0 * d)
    lv = [mosaic_px / 4.0 ** k for k in range(L + 1)]
    blend = lv[L] * (d + d_rgb) if L > 0 else lv[0] * (d + 7.0)
    for k in range(L, 0, -1):
        blend += lv[k] * d_rgb + lv[k - 1] * (d + (7.0 if k == 1 else d_rgb))
    return {"warp": warp, "feed": feed, "blend": blend, "total": warp + feed + blend}


class PairStitcher:
    """The n >= 2 tiles of one mosaic (normally a pair) -> one blended mosaic, buffers resident in HBM (torch CUDA
    tensors), on the deferred blender cycle (isx_blender_set_deferred_level0; more than 20 tiles: their descriptors in a device table)."""

    def __init__(self, imgs, K, Rs, scale, kind="cylindrical", num_bands=5, precision=_lib.PREC_F32,
                 device=0, stream=None, out_dtype="int16", deferred=True, interleave=False, verify_at=1, window=None, tile_type="u8"):
        """window = (x0, x1): this object produces only the columns [x0, x1) of the mosaic (x0 a multiple of _lib.WINDOW_GRANULE,
        counted from the mosaic's left edge) - one strip of a panorama cut across GPUs (mosaic.strip_windows).  It then warps and
        feeds only the tiles that can reach those columns (self.active; `imgs` may hold None for the others), prepare() still gets
        every tile's rectangle, and self.out is x1 - x0 columns wide: bit for bit the same columns as the unwindowed mosaic."""
        import torch
        self.torch = torch
        self.imgs, self.K, self.Rs = imgs, K, Rs
        self.device = device
        # tile_type "s16": the warped tiles are CV_16SC3 - the warp writes them so (the convertTo(CV_16S) of W:294 folded into its store) and
        # feed() receives what the reference's feed() receives (W:302); "u8": CV_8UC3 tiles through feed_u8 (the conversion fused into feed)
        self.tile_type = tile_type
        creator = CylindricalWarper if kind == "cylindrical" else SphericalWarper
        self.warper = creator(device, stream).create(scale)
        self.warper.set_deferred_verify(True)   # both ROI scans start after the last warp of a step (see step())
        self.blender = MultiBandBlender(False, num_bands, precision, device, stream)
        # the warped tiles and seam masks below live as long as this object: the deferred level-0 contract holds
        self.blender.set_deferred_level0(deferred)   # True: this object's buffers outlive blend(); "copy": feed() copies them
        # interleave: warp(t), feed(t), warp(t+1) ... with each tile's Gaussian chain on a side stream.  Measured
        # on MI355X: no gain (a kernel that fills every wave slot leaves nothing for a concurrent one), so off.
        self.interleave = interleave
        self.blender.set_overlap(self.interleave)
        self.precision, self.num_bands = precision, num_bands
        dev = torch.device("cuda", device)
        # This object works on `stream`; its inputs were produced, and its buffers are about to be allocated, on the caller's current stream.
        # torch's side streams are not ordered against that one: (1) the planning warps below must not read `imgs` before they exist, and
        # (2) a buffer torch.empty() hands out here may be a block the caller's stream has released but not finished with (the caching
        # allocator recycles within one stream's order only) - written from `stream` too early it corrupts whatever still reads that block.
        # (Round 5: bench.py's world-8 rehearsal found exactly that - the temporaries of its image generator recycled as warped-tile
        # buffers, one or two of 32 input images damaged for good on a busy GPU.)  So: order `stream` behind the caller's stream once.
        if stream is not None and hasattr(stream, "wait_stream"):
            stream.wait_stream(torch.cuda.current_stream(dev))
        # plan: ROI per tile (detectResultRoi), output buffers, seam masks
        src_size = next((im.shape[1], im.shape[0]) for im in imgs if im is not None)   # one rig: every tile has the same size
        self.rois = [self.warper.warpRoi(src_size if im is None else (im.shape[1], im.shape[0]), K, R) for im, R in zip(imgs, Rs)]
        self.sizes = [(r[2] - r[0] + 1, r[3] - r[1] + 1) for r in self.rois]
        self.corners = [(r[0], r[1]) for r in self.rois]
        self.window = None if window is None else (int(window[0]), int(window[1]))
        self.tile_cols = None
        self.active = list(range(len(imgs)))
        if self.window is not None:
            from . import mosaic
            self.active = mosaic.tiles_for_window(self.corners, self.sizes, num_bands, *self.window)
            if any(imgs[i] is None for i in self.active):
                raise ValueError("window %s needs tiles %s" % (self.window, self.active))
            # ... and of those tiles only the columns the strip depends on are warped in a step
            self.tile_cols = mosaic.tile_columns_for_window(self.corners, self.sizes, num_bands, *self.window)
        # The buffers this object writes from `stream` are ALLOCATED on `stream`: torch's caching allocator ties a block to the
        # stream it was allocated on and recycles it in that stream's order only - a buffer allocated on the caller's stream and written from
        # `stream` could be handed out again on the caller's stream, once this object is dropped, while `stream` still has writes queued.
        self._tstream = stream if isinstance(stream, torch.cuda.Stream) else None
        def empty(n, dtype):
            if isinstance(stream, torch.cuda.Stream):
                with torch.cuda.stream(stream):
                    return torch.empty((n,), dtype=dtype, device=dev)
            return torch.empty((n,), dtype=dtype, device=dev)
        # cv::Mat-style pitched buffers (row pitch a multiple of 64 B) so that the warp kernel can store dwords
        def pitched(h, row_bytes, shape, strides):
            pitch = (row_bytes + 63) // 64 * 64
            return empty(h * pitch, torch.uint8).as_strided(shape, (pitch,) + strides)
        act = set(self.active)
        if tile_type == "s16":
            def pitched16(h, w):
                pitch = (w * 6 + 63) // 64 * 64
                return empty(h * pitch // 2, torch.int16).as_strided((h, w, 3), (pitch // 2, 3, 1))
            self.warped = [pitched16(h, w) if i in act else None for i, (w, h) in enumerate(self.sizes)]
        else:
            self.warped = [pitched(h, w * 3, (h, w, 3), (3, 1)) if i in act else None for i, (w, h) in enumerate(self.sizes)]
        self.wmasks = [pitched(h, w, (h, w), (1,)) if i in act else None for i, (w, h) in enumerate(self.sizes)]
        for i in self.active:
            self.warper.warp_with_mask(imgs[i], K, Rs[i], dst_img=self.warped[i], dst_mask=self.wmasks[i])
        seam = synth.seam_masks(self.corners, [None if m is None else m.cpu().numpy() for m in self.wmasks], self.sizes)
        self.seam = [None if s is None else torch.from_numpy(s).to(dev) for s in seam]
        if isinstance(stream, torch.cuda.Stream):
            for t in self.seam:
                if t is not None:
                    t.record_stream(stream)     # uploaded on the caller's stream, read from `stream` for as long as this object lives
        self.roi_pad, (fw, fh), self.L = prepare_geometry(self.corners, self.sizes, num_bands)
        self.mosaic_size = (fw, fh)
        if self.window is not None:
            self.blender.set_window(*self.window)
            fw = self.window[1] - self.window[0]
        # Where the ROI verification scans of a planned step start.  verify_at < 0: right after the last warp (they then
        # run under the level-0 pyrDown, which they slow down by more than their own length).  verify_at = k >= 0:
        # inside blend(), behind the pyrDown launch of level k — from there to the last collapse step the launches are
        # small and leave most of the GPU idle.  Measured on MI355X, 4K pair: -1: 0.377 ms, 0: 0.371, 1: 0.354, 2: 0.362.
        # Round 3: a verification that is only a border scan (one workgroup; every spherical tile, and every cylindrical tile whose
        # extrema provably lie on its border) is not placed at all - it starts at once on the side stream with no event on the
        # main stream: 0.2167 -> 0.2100 ms (the event record behind the level-1 pyrDown was a 6 us bubble in the launch chain).
        self._verify_at_cfg = verify_at if deferred else None
        if verify_at is not None and verify_at >= 0 and all(
                self.warper.verify_is_light((self.imgs[i].shape[1], self.imgs[i].shape[0]), self.K, self.Rs[i]) for i in self.active):
            verify_at = -1
        if os.environ.get("ISX_VERIFY_AT", "") != "":
            verify_at = int(os.environ["ISX_VERIFY_AT"])
        self.mark = None
        if deferred and not interleave and verify_at is not None and verify_at >= 0 and self.L >= 1:
            self.mark = torch.cuda.Event()
            self.mark.record()   # materialise the hipEvent_t
            self.blender.set_mark_event(self.mark, min(verify_at, self.L - 1))
        odt = {"int16": torch.int16, "float32": torch.float32, "uint8": torch.uint8}[out_dtype]
        es = {"int16": 2, "float32": 4, "uint8": 1}[out_dtype]
        opitch = (fw * 3 * es + 63) // 64 * 64
        self.out = empty(fh * opitch // es, odt).as_strided((fh, fw, 3), (opitch // es, 3, 1))
        self.out_mask = pitched(fh, fw, (fh, fw), (1,))

    def _feed(self, i, corner):
        if self.tile_type == "s16":
            self.blender.feed(self.warped[i], self.seam[i], corner)
        else:
            self.blender.feed_u8(self.warped[i], self.seam[i], corner)

    def step(self):
        """Steady-state step without host round trips: the ROI scan (detectResultRoi) still runs on the
        GPU for every warp and is compared ON THE DEVICE with the ROI planned in __init__; a mismatch
        raises the sticky flag read by check_plan().  Everything else is the reference's call sequence."""
        if self.interleave:   # warp(t), feed(t), warp(t+1), ...: tile t's Gaussian chain runs under tile t+1's warp
            self.blender.prepare(self.corners, self.sizes)
            for i in self.active:
                self.warper.warp_with_mask_planned(self.imgs[i], self.K, self.Rs[i], self.rois[i], self.warped[i], self.wmasks[i])
                self._feed(i, self.corners[i])
        else:
            self._warps()
            if getattr(self, "_capturing_outside", False):
                self.warper.discard_pending()   # a step being captured: the verification runs beside the graph (replay)
            elif self.mark is None:
                self.warper.verify()   # the VALU-bound scans run on the side stream under the memory-bound pyramid kernels
            self.blender.prepare(self.corners, self.sizes)
            for i in self.active:
                self._feed(i, self.corners[i])
        self.blender.blend(self.out, self.out_mask)
        if self.mark is not None and not self.interleave:
            self.warper.verify_after(self.mark)   # blend() recorded the mark behind its level-`verify_at` pyrDown
        return self.out, self.out_mask

    def _warps(self):
        """The planned warps of every active tile - collected and launched as ONE kernel (isx_warper_begin_batch: blockIdx.z = tile; round 6.
        ISX_WARP_BATCH=0: one launch per tile, for A/B runs)."""
        batch = os.environ.get("ISX_WARP_BATCH", "1") != "0"
        if batch:
            self.warper.begin_batch()
        try:
            for i in self.active:
                if self.tile_cols is not None:
                    self.warper.set_dst_columns(*self.tile_cols[i])
                self.warper.warp_with_mask_planned(self.imgs[i], self.K, self.Rs[i], self.rois[i], self.warped[i], self.wmasks[i])
        finally:
            if self.tile_cols is not None:
                self.warper.set_dst_columns(0, 0)
            if batch:
                self.warper.end_batch()

    def step_until_blend(self):
        """The planned step up to (not including) blend(): warps, prepare, feeds - for step_batch."""
        self._warps()
        if getattr(self, "_capturing_outside", False):
            self.warper.discard_pending()
        elif self.mark is None:
            self.warper.verify()
        self.blender.prepare(self.corners, self.sizes)
        for i in self.active:
            self._feed(i, self.corners[i])

    @staticmethod
    def step_batch(stitchers):
        """The planned step of several independent mosaics with ONE chain of blend launches (isx_blender_blend_batch): every
        stitcher's warps and feeds as in step(), then one batched blend.  The stitchers share a stream."""
        from .blender import blend_batch
        for s in stitchers:
            s.step_until_blend()
        blend_batch([s.blender for s in stitchers], [s.out for s in stitchers], [s.out_mask for s in stitchers])
        for s in stitchers:
            if s.mark is not None and not s.interleave:
                s.warper.verify_after(s.mark)
        return [(s.out, s.out_mask) for s in stitchers]

    def capture(self):
        """Capture the planned step into a hipGraph (one launch per step instead of ~17 API calls):
        everything the step enqueues is stream work on resident buffers — no allocation, no host copy,
        no synchronisation — so it is capturable as is; the side-stream ROI scans are re-joined first."""
        torch = self.torch
        # A verification that is a border scan (self.mark is None: every tile's is) stays OUT of the graph: forked inside it by an event it cost
        # a replay 12 us (0.206 -> 0.218 ms at 4K), beside it nothing - replay() queues it from the rig alone and starts it on the
        # verification stream, as the eager step does.  A full-scan verification keeps its place inside (behind the level-`verify_at` pyrDown).
        # _verify_outside describes the GRAPH (replay() / verify_beside() read it); _capturing_outside is set only while the warm-up and the
        # captured step run, so that an eager step() after capture() verifies its plan inside the step again.
        self._verify_outside = self.mark is None and not self.interleave and os.environ.get("ISX_GRAPH_VERIFY_INSIDE", "") == ""
        self.gstream = torch.cuda.Stream(device=self.device)
        self.warper.set_stream(self.gstream)
        self.blender.set_stream(self.gstream)
        self.gstream.wait_stream(torch.cuda.current_stream(self.device))
        self._capturing_outside = self._verify_outside
        try:
            with torch.cuda.stream(self.gstream):
                self.step()
                self.warper.join()
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.gstream, capture_error_mode="relaxed"):
                self.step()
                if not self._verify_outside:
                    self.warper.join()      # (with the verification outside, the captured step never leaves its stream: nothing to join)
        finally:
            self._capturing_outside = False
        return self.graph

    def verify_beside(self):
        """The verification of a replayed step, beside the graph (see capture)."""
        if getattr(self, "_verify_outside", False):
            for i in self.active:
                self.warper.queue_verify((self.imgs[i].shape[1], self.imgs[i].shape[0]), self.K, self.Rs[i], self.rois[i])
            self.warper.verify()

    def replay(self):
        """Replays the captured step AND starts its plan's verification beside it.  A caller that replays the torch graph object directly
        (self.graph.replay()) must call verify_beside() itself, or check_plan() has nothing to report."""
        self.graph.replay()
        self.verify_beside()
        return self.out, self.out_mask

    @staticmethod
    def capture_batch(stitchers, branches=1, batch_size=0):
        """step_batch of several stitchers captured into ONE hipGraph (BASELINE config 3: a batch of independent pairs as a graph
        whose every launch spans all of them).  Returns (graph, stream); graph.replay() redoes the step of every stitcher, and
        verify_beside() of every stitcher then starts its plan's verification beside the graph (see capture).
        branches > 1 (round 6): the stitchers are dealt to that many PARALLEL chains inside the one graph - side streams forked from the capture
        stream by an event and joined back before the capture ends - so that one chain's launch-latency-bound small levels run under another's
        large kernels, as `bench.py --batch --streams 3` does eagerly (the single chain as a graph was level with the single eager chain and
        5 % behind three eager chains: VERDICT r5 item 6).  batch_size: mosaics per batched chain launch group (0: the whole branch)."""
        s0 = stitchers[0]
        torch = s0.torch
        branches = max(1, min(int(branches), len(stitchers)))
        gstream = torch.cuda.Stream(device=s0.device)
        streams = [gstream] + [torch.cuda.Stream(device=s0.device) for _ in range(branches - 1)]
        groups = [stitchers[g::branches] for g in range(branches)]
        for g, grp in enumerate(groups):
            for s in grp:
                s._verify_outside = s.mark is None and not s.interleave and os.environ.get("ISX_GRAPH_VERIFY_INSIDE", "") == ""   # see capture()
                s.gstream = gstream
                s.warper.set_stream(streams[g])
                s.blender.set_stream(streams[g])
        gstream.wait_stream(torch.cuda.current_stream(s0.device))

        def one(join_all):
            for g, grp in enumerate(groups):
                if g > 0:
                    streams[g].wait_stream(gstream)         # fork: the branch starts where the capture stream stands (under capture: a graph edge)
                with torch.cuda.stream(streams[g]):
                    n = len(grp) if batch_size <= 0 else batch_size
                    for q in range(0, len(grp), n):
                        PairStitcher.step_batch(grp[q:q + n])
                    for s in grp:
                        if join_all or not s._verify_outside:
                            s.warper.join()
            for g in range(1, branches):
                gstream.wait_stream(streams[g])             # join
        for s in stitchers:
            s._capturing_outside = s._verify_outside
        try:
            with torch.cuda.stream(gstream):
                one(True)
            torch.cuda.synchronize(s0.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=gstream, capture_error_mode="relaxed"):
                one(False)
        finally:
            for s in stitchers:
                s._capturing_outside = False
        for s in stitchers:
            s.graph = graph
            s._branch_streams = streams          # (kept alive with the graph)
        return graph, gstream

    def check_plan(self):
        """Synchronises; raises IsxError(ISX_ERR_PLAN) if any planned step saw a different ROI."""
        return self.warper.plan_status()

    def step_sync(self):
        """Exactly the reference's call sequence; the warper returns the corner to the host on every call
        (one stream synchronisation per tile), as cv::detail::RotationWarper::warp does."""
        cs = list(self.corners)   # the tiles this strip does not hold keep their planned corner
        for i in self.active:
            if self.tile_cols is not None:
                self.warper.set_dst_columns(*self.tile_cols[i])
            c, _, _ = self.warper.warp_with_mask(self.imgs[i], self.K, self.Rs[i], dst_img=self.warped[i], dst_mask=self.wmasks[i])
            cs[i] = c
        if self.tile_cols is not None:
            self.warper.set_dst_columns(0, 0)
        self.blender.prepare(cs, self.sizes)
        for i in self.active:
            self._feed(i, cs[i])
        self.blender.blend(self.out, self.out_mask)
        return self.out, self.out_mask

    def step_literal(self):
        """Call for call what a caller written against cv::detail::RotationWarper / cv::detail::Blender issues - the reference's main()
        through include/imagestitch_cv.hpp: per tile warp(img, K, R, INTER_LINEAR, BORDER_REFLECT) (W:229) and warp(mask, K, R, INTER_NEAREST,
        BORDER_CONSTANT) (W:232) of the all-255 source mask (W:213-214), EACH with its own detectResultRoi (isx_warper_roi, one host round
        trip) + isx_warper_warp_roi; convertTo(CV_16S) (W:294); prepare (W:281); feed(CV_16SC3, mask, corner) (W:302: construct with
        deferred="copy" so that feed() consumes its inputs); blend -> CV_16SC3 + mask (W:313).  Nothing fused, nothing planned."""
        from . import _lib as L
        from .blender import convert_to
        torch = self.torch
        if self.tile_type != "u8":
            raise ValueError("step_literal: the reference warps CV_8UC3 tiles (construct with tile_type='u8')")
        if not hasattr(self, "src_masks"):
            import contextlib
            with (torch.cuda.stream(self._tstream) if getattr(self, "_tstream", None) is not None else contextlib.nullcontext()):   # allocated on the stream that writes them
                # every mat as Mat::create makes it: continuous, rows back to back (a 3425-pixel CV_8UC3 row starts on an odd byte)
                self.src_masks = [None if im is None else torch.full(im.shape[:2], 255, dtype=torch.uint8, device=im.device) for im in self.imgs]   # W:213-214
                self.lit_warped = [None if wi is None else torch.empty(tuple(wi.shape), dtype=torch.uint8, device=wi.device) for wi in self.warped]
                self.lit_wmasks = [None if wm is None else torch.empty(tuple(wm.shape), dtype=torch.uint8, device=wm.device) for wm in self.wmasks]
                self.warped16 = [None if wi is None else torch.empty(tuple(wi.shape), dtype=torch.int16, device=wi.device) for wi in self.warped]
                self.lit_out = torch.empty(tuple(self.out.shape), dtype=self.out.dtype, device=self.out.device)
                self.lit_out_mask = torch.empty(tuple(self.out_mask.shape), dtype=torch.uint8, device=self.out.device)
        cs = list(self.corners)
        for i in self.active:
            im = self.imgs[i]
            size = (im.shape[1], im.shape[0])
            roi = self.warper.warpRoi(size, self.K, self.Rs[i])                                                                      # W:126 inside W:229
            self.warper.warp_roi(im, self.K, self.Rs[i], L.INTER_LINEAR, L.BORDER_REFLECT, roi, self.lit_warped[i])
            roi = self.warper.warpRoi(size, self.K, self.Rs[i])                                                                      # W:126 inside W:232
            self.warper.warp_roi(self.src_masks[i], self.K, self.Rs[i], L.INTER_NEAREST, L.BORDER_CONSTANT, roi, self.lit_wmasks[i])
            cs[i] = (roi[0], roi[1])
            convert_to(self.lit_warped[i], np.int16, dst=self.warped16[i], device=self.device, stream=self.blender._stream_obj)        # W:294
        self.blender.prepare(cs, self.sizes)                                                                                         # W:281
        for i in self.active:
            self.blender.feed(self.warped16[i], self.seam[i], cs[i])                                                                 # W:302
        self.blender.blend(self.lit_out, self.lit_out_mask)                                                                          # W:313
        return self.lit_out, self.lit_out_mask

    def bytes_model(self):
        src_px = [self.imgs[i].shape[0] * self.imgs[i].shape[1] for i in self.active]
        warped_px = [self.sizes[i][0] * self.sizes[i][1] for i in self.active]
        base = [np.prod(feed_geometry(self.roi_pad, self.L, self.corners[i], self.sizes[i])) for i in self.active]
        mosaic = self.roi_pad[2] * self.roi_pad[3]
        out = model_bytes(src_px, warped_px, [float(b) for b in base], float(mosaic), self.precision, self.L)
        out.update({"src_px": src_px, "warped_px": warped_px, "tile_base_px": [int(b) for b in base], "mosaic_px": int(mosaic)})
        return out


MosaicStitcher = PairStitcher   # the same object under the name that fits n > 2 tiles (BASELINE config 5: a row of 8)


class SplitStitcher:
    """ONE mosaic computed as `nsplit` column strips on `nsplit` streams of one GPU, the strips' launch chains staggered: a step of one
    mosaic is about a dozen dependent launches of which six (pyramid levels 2 and up) are latency-bound and leave the GPU nearly idle
    (39 of 211 us at 4K); a strip is bit for bit the same columns of the whole blend (isx_blender_set_window: recomputed halos,
    no exchange), so the strips are independent chains, and strip i + 1 starts when strip i has issued its two full-size pyramid kernels
    (the event isx_blender_set_mark_event records behind the level-1 pyrDown) - its large kernels then run beside strip i's small ones.
    Every strip warps only the tile columns it needs and writes its columns of ONE output mat.  Within a step only: the first strip of a
    step waits for the last strip of the step before.
    MEASURED (tools/probes/split_probe.py, profiles/round4_split_strips.txt): identical mosaics, and SLOWER than the single chain - two strips
    are two chains of launches, the caller's thread needs 0.1 ms to enqueue one chain, and the step becomes bound by that (0.28 - 0.37 ms
    against 0.21).  Kept as the record of that experiment (and as a user of the window machinery on one GPU), not used by bench.py."""

    def __init__(self, imgs, K, Rs, scale, kind="cylindrical", num_bands=5, precision=_lib.PREC_F32, device=0, out_dtype="int16",
                 nsplit=2, tile_type="u8", stagger_level=1, chain_steps=False):
        import torch
        from . import mosaic
        self.torch = torch
        dev = torch.device("cuda", device)
        creator = CylindricalWarper if kind == "cylindrical" else SphericalWarper
        wp = creator(device, None).create(scale)
        src_size = (imgs[0].shape[1], imgs[0].shape[0])
        rois = [wp.warpRoi(src_size, K, R) for R in Rs]
        del wp
        corners = [(r[0], r[1]) for r in rois]
        sizes = [(r[2] - r[0] + 1, r[3] - r[1] + 1) for r in rois]
        _, (fw, fh), _ = prepare_geometry(corners, sizes, num_bands)
        windows, sw = mosaic.strip_windows(fw, nsplit, _lib.WINDOW_GRANULE)
        windows = [w for w in windows if w[1] > w[0]]
        self.mosaic_size = (fw, fh)
        odt = {"int16": torch.int16, "float32": torch.float32, "uint8": torch.uint8}[out_dtype]
        es = {"int16": 2, "float32": 4, "uint8": 1}[out_dtype]
        wpad = len(windows) * sw                       # the last strip may reach past the mosaic's right edge: columns nobody writes
        opitch = (wpad * 3 * es + 127) // 128 * 128
        self._out_full = torch.empty((fh * opitch // es,), dtype=odt, device=dev).as_strided((fh, wpad, 3), (opitch // es, 3, 1))
        mpitch = (wpad + 127) // 128 * 128
        self._mask_full = torch.empty((fh * mpitch,), dtype=torch.uint8, device=dev).as_strided((fh, wpad), (mpitch, 1))
        self.out, self.out_mask = self._out_full[:, :fw], self._mask_full[:, :fw]
        self.streams = [torch.cuda.Stream(device=device) for _ in windows]
        self.parts = []
        for w, st in zip(windows, self.streams):
            part = PairStitcher(imgs, K, Rs, scale, kind, num_bands, precision, device, st, out_dtype, deferred=True, window=w, tile_type=tile_type)
            part.out = self._out_full[:, w[0]:w[1]]
            part.out_mask = self._mask_full[:, w[0]:w[1]]
            self.parts.append(part)
        torch.cuda.synchronize(device)
        self.corners, self.sizes = self.parts[0].corners, self.parts[0].sizes
        # the stagger: strip i records `go[i]` behind its level-`stagger_level` pyrDown, strip i + 1's chain starts there.  (A part whose ROI
        # verification is a full scan already uses the blender's one mark event to place it: that part is then simply not staggered.)
        self.go = []
        for part in self.parts[:-1]:
            ev = None
            if part.mark is None and part.L >= 1 and stagger_level is not None:
                ev = torch.cuda.Event()
                ev.record()
                part.blender.set_mark_event(ev, min(stagger_level, part.L - 1))
            self.go.append(ev)
        self.done = [torch.cuda.Event() for _ in self.parts]
        self.chain_steps = chain_steps
        self._first = True

    def step(self):
        torch = self.torch
        main = torch.cuda.current_stream()
        for i, (part, st) in enumerate(zip(self.parts, self.streams)):
            st.wait_stream(main)                                   # whatever produced the sources
            if i == 0 and not self._first and not self.chain_steps:
                st.wait_event(self.done[-1])                       # one step at a time: the last strip of the step before has finished
            elif i > 0 and self.go[i - 1] is not None:
                pass                                               # (the wait is enqueued below, once strip i - 1's step() has recorded the event)
            with torch.cuda.stream(st):
                if i > 0 and self.go[i - 1] is not None:
                    st.wait_event(self.go[i - 1])
                part.step()
                self.done[i].record(st)
        for ev in self.done:
            main.wait_event(ev)
        self._first = False
        return self.out, self.out_mask

    def check_plan(self):
        return sum(p.check_plan() for p in self.parts)

