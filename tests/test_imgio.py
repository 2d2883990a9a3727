"""the .bmp reader / writer either side of the path (cv::imread W:166, cv::imwrite W:155-156,315),
against Pillow's codec.  Host-only code: runs without a GPU."""
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def isx():
    import imagestitch_amd
    imagestitch_amd.load()
    return imagestitch_amd


@pytest.mark.parametrize("shape", [(7, 5), (33, 64), (20, 101), (1, 1), (64, 3)])
def test_bmp_round_trip_against_pillow(isx, tmp_path, shape):
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    h, w = shape
    bgr = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    gray = rng.integers(0, 256, (h, w)).astype(np.uint8)
    p = str(tmp_path / "a.bmp")
    assert isx.imwrite(p, bgr)
    assert np.array_equal(np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1], bgr)           # our writer, Pillow's reader
    assert np.array_equal(isx.imread(p), bgr)                                                  # our reader
    PIL.fromarray(bgr[:, :, ::-1].copy()).save(str(tmp_path / "b.bmp"))                        # Pillow's writer, our reader
    assert np.array_equal(isx.imread(str(tmp_path / "b.bmp")), bgr)
    # CV_8UC1 -> 8-bit bitmap with a grey palette; imread (IMREAD_COLOR) expands it to 3 equal channels
    assert isx.imwrite(str(tmp_path / "g.bmp"), gray)
    im = PIL.open(str(tmp_path / "g.bmp"))
    assert im.mode in ("L", "P") and np.array_equal(np.asarray(im.convert("L")), gray)
    assert np.array_equal(isx.imread(str(tmp_path / "g.bmp")), np.repeat(gray[:, :, None], 3, 2))
    PIL.fromarray(gray).save(str(tmp_path / "h.bmp"))
    assert np.array_equal(isx.imread(str(tmp_path / "h.bmp")), np.repeat(gray[:, :, None], 3, 2))
    # a pitched (non-dense) host view is written row by row
    big = rng.integers(0, 256, (h, w + 5, 3)).astype(np.uint8)
    assert isx.imwrite(str(tmp_path / "v.bmp"), big[:, 2:2 + w])
    assert np.array_equal(isx.imread(str(tmp_path / "v.bmp")), big[:, 2:2 + w])


def test_bmp_top_down_and_errors(isx, tmp_path):
    bgr = np.arange(4 * 3 * 3, dtype=np.uint8).reshape(4, 3, 3)
    p = str(tmp_path / "t.bmp")
    isx.imwrite(p, bgr)
    raw = bytearray(open(p, "rb").read())
    raw[22:26] = (-4 & 0xffffffff).to_bytes(4, "little")          # negative height = top-down rows
    open(p, "wb").write(bytes(raw))
    assert np.array_equal(isx.imread(p), bgr[::-1])
    with pytest.raises(isx.IsxError):
        isx.imread(str(tmp_path / "missing.bmp"))
    open(str(tmp_path / "x.bmp"), "wb").write(b"\x89PNG\r\n\x1a\n" + b"\0" * 100)          # a PNG signature: neither decoder takes it
    with pytest.raises(isx.IsxError) as e:
        isx.imread(str(tmp_path / "x.bmp"))
    assert e.value.code == 6
    with pytest.raises(isx.IsxError):
        isx.imwrite(str(tmp_path / "f.bmp"), np.zeros((4, 4, 3), np.float32))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree only exists in the build container")
def test_reads_the_references_committed_bitmaps(isx):
    """The artefacts the reference's imwrite produced (S:1195-1198) decode to what Pillow decodes."""
    import glob
    files = sorted(glob.glob("/root/reference/**/*.bmp", recursive=True))
    assert files
    for f in files[:6]:
        ref = np.asarray(PIL.open(f).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(isx.imread(f), ref), f


def test_write_mosaics_of_a_gathered_batch(isx, tmp_path):
    """The tiled-mosaic writer on the (world, capacity) tensor the all-gather produces (host tensors here)."""
    import torch
    from imagestitch_amd import mosaic
    rng = np.random.default_rng(9)
    shapes = [[(5, 7, 3), (6, 4, 3)], [(3, 9, 3)]]
    imgs = [[torch.from_numpy(rng.integers(0, 256, s).astype(np.uint8)) for s in row] for row in shapes]
    cap = max(sum(t.numel() for t in row) for row in imgs)
    gathered = torch.stack([mosaic.pack_blocks(row, cap) for row in imgs])
    names = mosaic.write_mosaics(str(tmp_path / "pano"), gathered, shapes)
    assert [os.path.basename(n) for n in names] == ["pano_r0_p0.bmp", "pano_r0_p1.bmp", "pano_r1_p0.bmp"]
    flat = [t for row in imgs for t in row]
    for n, t in zip(names, flat):
        assert np.array_equal(isx.imread(n), t.numpy())


@pytest.mark.gpu
def test_bmp_device_mats(isx, tmp_path):
    import torch
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    t = torch.from_numpy(a).cuda()
    p = str(tmp_path / "d.bmp")
    isx.imwrite(p, t)
    assert np.array_equal(np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1], a)
    back = isx.imread(p, device=0)
    assert back.is_cuda and np.array_equal(back.cpu().numpy(), a)
    pitched = torch.zeros((37, 64, 3), dtype=torch.uint8, device="cuda")
    pitched[:, 3:56] = t
    isx.imwrite(p, pitched[:, 3:56])
    assert np.array_equal(isx.imread(p), a)


def _psnr(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).mean()
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 ** 2 / mse)


def _smooth(h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([128 + 100 * np.sin(xx / 23.0) * np.cos(yy / 17.0), 128 + 90 * np.cos(xx / 31.0 + 1), 60 + xx * 0.5 + yy * 0.2], 2).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(203, 317), (16, 16), (1, 1), (7, 33), (64, 9)])
def test_jpeg_write_is_read_by_a_stock_decoder(isx, tmp_path, size):
    """imwrite("pano.jpg", result) (S:1282): baseline JFIF at OpenCV's defaults (quality 95, 4:2:0).  The file is decoded
    with Pillow's libjpeg; its error against the source is that of libjpeg's own encoder at the same settings."""
    Image = PIL
    h, w = size
    img = _smooth(h, w)
    p = str(tmp_path / "t.jpg")
    assert isx.imwrite(p, img)
    with Image.open(p) as im:
        assert im.format == "JPEG" and im.size == (w, h) and im.mode == "RGB"
        dec = np.asarray(im.convert("RGB"))[:, :, ::-1]
    ref = str(tmp_path / "ref.jpg")
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(ref, quality=95, subsampling=2)
    with Image.open(ref) as im:
        dec_ref = np.asarray(im.convert("RGB"))[:, :, ::-1]
    assert _psnr(dec, img) > min(40.0, _psnr(dec_ref, img) - 1.5), (_psnr(dec, img), _psnr(dec_ref, img))


def test_jpeg_grey_quality_and_errors(isx, tmp_path):
    Image = PIL
    img = _smooth(120, 150)
    g = np.ascontiguousarray(img[:, :, 0])
    p = str(tmp_path / "g.jpeg")
    isx.imwrite(p, g, quality=90)
    with Image.open(p) as im:
        assert im.mode == "L" and im.size == (150, 120)
        assert _psnr(np.asarray(im), g) > 45.0
    sizes, psnrs = [], []
    for q in (10, 50, 95):
        pq = str(tmp_path / ("q%d.jpg" % q))
        isx.imwrite(pq, img, quality=q)
        with Image.open(pq) as im:
            psnrs.append(_psnr(np.asarray(im.convert("RGB"))[:, :, ::-1], img))
        sizes.append(os.path.getsize(pq))
    assert sizes[0] < sizes[1] < sizes[2] and psnrs[0] < psnrs[1] < psnrs[2]
    # noise: every Huffman symbol class, long runs of 0xff bytes in the stream (byte stuffing)
    noise = np.random.default_rng(5).integers(0, 256, (97, 131, 3), dtype=np.uint8)
    pn = str(tmp_path / "n.jpg")
    isx.imwrite(pn, noise, quality=100)
    with Image.open(pn) as im:
        im.load()
        assert im.size == (131, 97)
    with pytest.raises(isx.IsxError):
        isx.imwrite(str(tmp_path / "bad.jpg"), img, quality=0)
    with pytest.raises(isx.IsxError):
        isx.imwrite(str(tmp_path / "f.jpg"), img.astype(np.float32))


@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (1, 1), (2, 3), (17, 9), (100, 131), (33, 3)])
def test_jpeg_read_equals_libjpeg_bit_for_bit(isx, tmp_path, shape):
    """cv::imread of a .jpg is libjpeg's decode (accurate integer IDCT, fancy upsampling, fixed-point YCbCr -> RGB): the decoder of
    csrc/jpegdec.cpp against Pillow's libjpeg-turbo on 4:4:4 / 4:2:2 / 4:2:0 / grey files, three qualities, with and without restart
    intervals, noise and smooth content."""
    from imagestitch_amd import synth
    rng = np.random.default_rng(shape[0] * 977 + shape[1])
    h, w = shape
    noise = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    smooth = synth.make_tile(max(h, 8), max(w, 8), 5)[:h, :w].copy()
    p = str(tmp_path / "t.jpg")
    for src in (noise, smooth):
        for sub in (0, 1, 2):
            for q in (30, 75, 95):
                for ri in (0, 3):
                    kw = dict(quality=q, subsampling=sub)
                    if ri:
                        kw["restart_marker_blocks"] = ri
                    PIL.fromarray(src).save(p, "JPEG", **kw)
                    ref = np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1]
                    assert np.array_equal(isx.imread(p), ref), (shape, sub, q, ri)
    g = rng.integers(0, 256, (h, w)).astype(np.uint8)
    PIL.fromarray(g).save(p, "JPEG", quality=80)
    assert np.array_equal(isx.imread(p), np.repeat(np.asarray(PIL.open(p))[:, :, None], 3, 2))


@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (1, 1), (2, 3), (17, 9), (100, 131), (33, 3), (240, 321)])
def test_progressive_jpeg_read_equals_libjpeg_bit_for_bit(isx, tmp_path, shape):
    """Progressive files (SOF2: spectral selection + successive approximation, DC / AC first and refinement scans, end-of-band runs) -
    what Pillow's libjpeg-turbo writes with progressive=True, with and without optimised tables and restart intervals, 4:4:4 / 4:2:2 /
    4:2:0 / grey - decode to the bytes libjpeg returns."""
    from imagestitch_amd import synth
    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    h, w = shape
    noise = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    smooth = synth.make_tile(max(h, 8), max(w, 8), 6)[:h, :w].copy()
    flat = np.full((h, w, 3), 200, np.uint8); flat[h // 2:, :, 1] = 17           # long end-of-band runs
    p = str(tmp_path / "p.jpg")
    n = 0
    from PIL import ImageFile
    ImageFile.MAXBLOCK = max(ImageFile.MAXBLOCK, 8 * h * w + 65536)      # Pillow's encoder cannot suspend inside a progressive scan: give it the whole file
    for src in (noise, smooth, flat):
        for sub in (0, 1, 2):
            for q in (25, 75, 97):
                for kw in (dict(), dict(optimize=True), dict(restart_marker_blocks=2), dict(restart_marker_rows=1)):
                    PIL.fromarray(src).save(p, "JPEG", quality=q, subsampling=sub, progressive=True, **kw)
                    assert b"\xff\xc2" in open(p, "rb").read()
                    ref = np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1]
                    assert np.array_equal(isx.imread(p), ref), (shape, sub, q, kw)
                    n += 1
    g = rng.integers(0, 256, (h, w)).astype(np.uint8)
    PIL.fromarray(g).save(p, "JPEG", quality=80, progressive=True)
    assert np.array_equal(isx.imread(p), np.repeat(np.asarray(PIL.open(p))[:, :, None], 3, 2))
    assert n == 108


def _one_bit_per_block_progressive_jpeg(size):
    """A valid grey SOF2 file that spends ONE bit per block: a DC-first scan of zero differences under a 1-bit Huffman code and one AC scan of
    end-of-band runs (what a mozjpeg-style scan script without DC successive approximation makes of a flat image)."""
    import struct
    def seg(marker, payload):
        return bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload
    blocks = (size // 8) ** 2
    assert size % 8 == 0 and blocks % 16384 == 0
    out = b"\xff\xd8"
    out += seg(0xDB, bytes([0]) + bytes([16] * 64))                                       # DQT 0
    out += seg(0xC2, bytes([8]) + struct.pack(">HH", size, size) + bytes([1, 1, 0x11, 0]))   # SOF2: one component, 1 x 1, table 0
    out += seg(0xC4, bytes([0x00]) + bytes([1] + [0] * 15) + bytes([0]))                  # DC table 0: the code "0" = category 0
    out += seg(0xDA, bytes([1, 1, 0x00, 0, 0, 0x00]))                                     # SOS: DC first, Ss = Se = 0, Ah = Al = 0
    out += bytes(blocks // 8)                                                             # one "0" bit per block
    out += seg(0xC4, bytes([0x10]) + bytes([1] + [0] * 15) + bytes([0xE0]))               # AC table 0: the code "0" = EOB14 (a run of 16384 + 14 extra bits)
    out += seg(0xDA, bytes([1, 1, 0x00, 1, 63, 0x00]))                                    # SOS: AC first, 1..63, Ah = Al = 0
    nbits = 15 * (blocks // 16384)
    pad = (-nbits) % 8
    out += int(("0" * nbits + "1" * pad), 2).to_bytes((nbits + pad) // 8, "big")
    return out + b"\xff\xd9"


def test_progressive_jpeg_of_one_bit_per_block_is_read(isx, tmp_path):
    """the frame-size plausibility bound (two bits per block) holds for sequential files only.  2048 x 2048 in 8.3 KB."""
    data = _one_bit_per_block_progressive_jpeg(2048)
    assert len(data) < 8400
    p = str(tmp_path / "flat.jpg")
    open(p, "wb").write(data)
    ref = np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1]
    assert ref.shape == (2048, 2048, 3) and (ref == 128).all()
    assert np.array_equal(isx.imread(p), ref)
    # the sequential bound is still there: the same header as SOF0 over the same few bytes is refused before anything is allocated
    bad = data.replace(b"\xff\xc2", b"\xff\xc0", 1)
    open(p, "wb").write(bad)
    with pytest.raises(isx.IsxError):
        isx.imread(p)


def test_progressive_jpeg_incomplete_scans_are_refused(isx, tmp_path):
    """A progressive file cut before its last scans: libjpeg would decode what it has and smooth the blocks; this reader says so instead
    of returning different bytes."""
    from imagestitch_amd import synth, IsxError
    img = synth.make_tile(96, 128, 4)
    p = str(tmp_path / "p.jpg")
    PIL.fromarray(img).save(p, "JPEG", quality=85, progressive=True)
    raw = open(p, "rb").read()
    scans = [i for i in range(len(raw) - 1) if raw[i] == 0xFF and raw[i + 1] == 0xDA]
    assert len(scans) >= 6
    open(str(tmp_path / "cut.jpg"), "wb").write(raw[:scans[-1]] + b"\xff\xd9")
    with pytest.raises(IsxError) as e:
        isx.imread(str(tmp_path / "cut.jpg"))
    assert e.value.code == 4 or "progressive" in str(e.value)


def test_jpeg_write_then_read_and_unsupported_files(isx, tmp_path):
    from imagestitch_amd import synth, IsxError
    img = synth.make_tile(120, 200, 9)
    p = str(tmp_path / "w.jpg")
    assert isx.imwrite(p, img)                                                  # the library's own JFIF writer ...
    ref = np.asarray(PIL.open(p).convert("RGB"))[:, :, ::-1]
    got = isx.imread(p)                                                          # ... read back by its own decoder = libjpeg's decode of that file
    assert np.array_equal(got, ref)
    assert np.abs(got.astype(int) - img.astype(int)).mean() < 16.0             # (+-32 noise per pixel, 4:2:0 chroma at quality 95)
    PIL.fromarray(img[:, :, ::-1].copy()).save(str(tmp_path / "base.jpg"), "JPEG", quality=80)
    raw = bytearray(open(str(tmp_path / "base.jpg"), "rb").read())
    i = raw.index(b"\xff\xc0")
    raw[i + 1] = 0xC9                                                          # the frame header of an arithmetic-coded file: not decoded
    open(str(tmp_path / "arith.jpg"), "wb").write(bytes(raw))
    with pytest.raises(IsxError):
        isx.imread(str(tmp_path / "arith.jpg"))
    open(str(tmp_path / "junk.jpg"), "wb").write(b"\xff\xd8\xff\xdb\x00")
    with pytest.raises(IsxError):
        isx.imread(str(tmp_path / "junk.jpg"))


def test_jpeg_read_of_the_references_own_panorama(isx):
    """pano.jpg (S:1282) as the reference committed it, decoded as cv::imread would: equal to libjpeg's decode.  Build container only."""
    import glob
    files = sorted(glob.glob("/root/reference/*/*/pano.jpg"))
    if not files:
        pytest.skip("reference tree not present")
    for f in files[:2]:
        ref = np.asarray(PIL.open(f).convert("RGB"))[:, :, ::-1]
        assert np.array_equal(isx.imread(f), ref), f


def test_jpeg_read_survives_corrupted_files(isx, tmp_path):
    """600 damaged files (flipped bytes, truncations, deletions, insertions, damaged headers; baseline, grey and progressive
    originals): every one is either decoded into a 3-channel image or refused with an error - the decoder never reads or writes
    out of bounds (the same corpus runs clean under AddressSanitizer / UBSan, tools/probes/jpeg_asan.sh)."""
    import io
    PIL = pytest.importorskip("PIL.Image")
    from imagestitch_amd._lib import IsxError
    rng = np.random.default_rng(9)
    base = []
    for (w, h, q, sub) in [(37, 29, 75, 2), (64, 48, 90, 0), (100, 131, 50, 1), (17, 9, 95, 2), (1, 1, 90, 2), (3, 200, 70, 1)]:
        b = io.BytesIO()
        PIL.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(b, "JPEG", quality=q, subsampling=sub)
        base.append(b.getvalue())
    b = io.BytesIO(); PIL.fromarray(rng.integers(0, 255, (40, 50), dtype=np.uint8)).save(b, "JPEG", quality=80); base.append(b.getvalue())
    b = io.BytesIO(); PIL.fromarray(rng.integers(0, 255, (40, 50, 3), dtype=np.uint8)).save(b, "JPEG", quality=80, progressive=True); base.append(b.getvalue())
    decoded = refused = 0
    path = str(tmp_path / "damaged.jpg")
    for it in range(600):
        data = bytearray(base[it % len(base)])
        mode = it % 5
        if mode == 0:
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(2, len(data)))] = int(rng.integers(0, 256))
        elif mode == 1:
            data = data[:int(rng.integers(2, len(data)))]
        elif mode == 2:
            q = int(rng.integers(2, len(data))); del data[q:q + int(rng.integers(1, 40))]
        elif mode == 3:
            q = int(rng.integers(2, len(data))); data[q:q] = bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8))
        else:
            for _ in range(3):
                data[int(rng.integers(2, min(len(data), 200)))] = int(rng.integers(0, 256))
        with open(path, "wb") as f:
            f.write(bytes(data))
        try:
            im = isx.imread(path)
            assert im.ndim == 3 and im.shape[2] == 3 and im.dtype == np.uint8
            decoded += 1
        except IsxError:
            refused += 1
    assert decoded > 50 and refused > 50


def _jpeg_segments(data):
    """Split a baseline JPEG into (marker, payload-with-length) segments up to and including SOS + its entropy-coded data."""
    segs, p = [], 2
    while p < len(data):
        assert data[p] == 0xFF
        m = data[p + 1]
        n = (data[p + 2] << 8) | data[p + 3]
        if m == 0xDA:
            segs.append((m, bytes(data[p:])))      # scan header + entropy-coded data + EOI
            break
        segs.append((m, bytes(data[p:p + 2 + n])))
        p += 2 + n
    return segs


def test_jpeg_read_refuses_structurally_hostile_files(isx, tmp_path):
    """Structures byte-level mutation cannot produce: a second frame header that enlarges the image after the coefficient
    buffers were sized; DRI / SOS segments that end before their payload, at the very end of the file; a few-hundred-byte file that
    declares 65535 x 65535 pixels.  Each is refused with an error (the first used to write past the heap, the last to throw bad_alloc)."""
    import io
    from imagestitch_amd._lib import IsxError
    rng = np.random.default_rng(2)

    def jpeg(w, h):
        b = io.BytesIO()
        PIL.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(b, "JPEG", quality=80, subsampling=0)
        return b.getvalue()
    small, big = jpeg(8, 8), jpeg(512, 512)
    path = str(tmp_path / "hostile.jpg")

    def refused(data, what):
        with open(path, "wb") as f:
            f.write(data)
        with pytest.raises(IsxError):
            isx.imread(path)
        # the C entry point itself, with an output mat of the size the FIRST header declares (the caller of a C-ABI sizes its own buffer)
        import ctypes as C
        from imagestitch_amd import _lib
        out = np.zeros((8, 8, 3), np.uint8)
        m = _lib.as_mat(out)
        assert _lib.load().isx_jpeg_read(path.encode(), C.byref(m)) != 0, what

    s_small, s_big = _jpeg_segments(small), _jpeg_segments(big)
    sof_big = [seg for mk, seg in s_big if mk == 0xC0][0]
    sos_big = [seg for mk, seg in s_big if mk == 0xDA][0]
    # 1. 8 x 8 frame + its scan, then a 512 x 512 SOF + scan
    body = b"".join(seg for _, seg in s_small)
    assert body.endswith(b"\xff\xd9")
    refused(b"\xff\xd8" + body[:-2] + sof_big + sos_big, "second SOF")
    # 2. a marker segment of length 2 that ends the file: DRI, SOS
    head = b"".join(seg for mk, seg in s_small if mk != 0xDA)
    refused(b"\xff\xd8" + head + b"\xff\xdd\x00\x02", "DRI without payload")
    refused(b"\xff\xd8" + head + b"\xff\xda\x00\x02", "SOS without payload")
    # 3. huge declared size in a tiny file
    sof = bytearray([seg for mk, seg in s_small if mk == 0xC0][0])
    sof[5:9] = b"\xff\xff\xff\xff"
    parts = [bytes(sof) if mk == 0xC0 else seg for mk, seg in s_small]
    refused(b"\xff\xd8" + b"".join(parts), "65535 x 65535 in a few hundred bytes")
    # the untouched files still decode
    with open(path, "wb") as f:
        f.write(big)
    assert isx.imread(path).shape == (512, 512, 3)
