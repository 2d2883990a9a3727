#!/usr/bin/env bash
# The JPEG reader under AddressSanitizer + UBSan on a corpus of damaged files (host code only; no GPU needed):
#   bash tools/probes/jpeg_asan.sh            -> "ok N bad M" and no sanitizer report
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d); cd "$W"
cat > harness.cpp <<'CPP'
#include <cstdio>
#include <vector>
#include "imagestitch_hip.h"
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    for (int i = 1; i < argc; ++i) {
        int r = 0, c = 0;
        if (isx_jpeg_size(argv[i], &r, &c) != ISX_OK || r <= 0 || c <= 0 || (long long)r * c > 4000000) { ++bad; continue; }
        std::vector<unsigned char> buf((size_t)r * c * 3);
        isx_mat m; m.data = buf.data(); m.rows = r; m.cols = c; m.type = ISX_8UC3; m.step = (size_t)c * 3; m.device = -1;
        if (isx_jpeg_read(argv[i], &m) == ISX_OK) ++ok; else ++bad;
    }
    printf("ok %d bad %d\n", ok, bad);
    return 0;
}
CPP
python3 - <<'PY'
import io, os, numpy as np
from PIL import Image
rng = np.random.default_rng(9)
base = []
for (w, h, q, sub) in [(37, 29, 75, 2), (64, 48, 90, 0), (100, 131, 50, 1), (17, 9, 95, 2), (33, 33, 85, 2), (8, 8, 60, 0), (1, 1, 90, 2), (3, 200, 70, 1)]:
    b = io.BytesIO(); Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(b, "JPEG", quality=q, subsampling=sub); base.append(b.getvalue())
g = io.BytesIO(); Image.fromarray(rng.integers(0, 255, (40, 50), dtype=np.uint8)).save(g, "JPEG", quality=80); base.append(g.getvalue())
p = io.BytesIO(); Image.fromarray(rng.integers(0, 255, (40, 50, 3), dtype=np.uint8)).save(p, "JPEG", quality=80, progressive=True); base.append(p.getvalue())
for (w, h, q, sub, kw) in [(61, 47, 70, 2, {}), (90, 33, 92, 0, {"optimize": True}), (40, 77, 40, 1, {"restart_marker_blocks": 2}), (24, 24, 85, 2, {})]:      # progressive: half of the corpus
    for _ in range(2):
        b = io.BytesIO(); Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(b, "JPEG", quality=q, subsampling=sub, progressive=True, **kw); base.append(b.getvalue())
os.makedirs("c", exist_ok=True)
for it in range(4000):
    data = bytearray(base[it % len(base)]); mode = it % 5
    if mode == 0:
        for _ in range(int(rng.integers(1, 6))): data[int(rng.integers(2, len(data)))] = int(rng.integers(0, 256))
    elif mode == 1: data = data[:int(rng.integers(2, len(data)))]
    elif mode == 2:
        q = int(rng.integers(2, len(data))); del data[q:q + int(rng.integers(1, 40))]
    elif mode == 3:
        q = int(rng.integers(2, len(data))); data[q:q] = bytes(rng.integers(0, 256, int(rng.integers(1, 30)), dtype=np.uint8))
    else:
        for _ in range(3): data[int(rng.integers(2, min(len(data), 200)))] = int(rng.integers(0, 256))
    open("c/%04d.jpg" % it, "wb").write(bytes(data))
# structures no byte-level mutation produces: a second, larger frame header after a scan; DRI / SOS segments that end before
# their payload at the end of the file; 65535 x 65535 declared by a few hundred bytes; every pair of originals spliced header-to-scan
def segs(d):
    out, p = [], 2
    while p < len(d):
        m = d[p + 1]; n = (d[p + 2] << 8) | d[p + 3]
        if m == 0xDA: out.append((m, bytes(d[p:]))); break
        out.append((m, bytes(d[p:p + 2 + n]))); p += 2 + n
    return out
k = 0
for a in base[:9]:
    sa = segs(a)
    head = b"".join(s for m, s in sa if m != 0xDA); body = b"".join(s for m, s in sa)
    for tail in (b"\xff\xdd\x00\x02", b"\xff\xda\x00\x02", b"\xff\xc4\x00\x02", b"\xff\xdb\x00\x02", b"\xff\xc0\x00\x02"):
        open("c/s%04d.jpg" % k, "wb").write(b"\xff\xd8" + head + tail); k += 1
    for b2 in base[:9]:
        sb = segs(b2)
        extra = b"".join(s for m, s in sb if m in (0xC0, 0xDA))
        open("c/s%04d.jpg" % k, "wb").write(b"\xff\xd8" + body[:-2] + extra); k += 1
    sof = bytearray([s for m, s in sa if m == 0xC0][0]); sof[5:9] = b"\xff\xff\xff\xff"
    open("c/s%04d.jpg" % k, "wb").write(b"\xff\xd8" + b"".join(bytes(sof) if m == 0xC0 else s for m, s in sa)); k += 1
PY
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer --offload-arch=gfx950 -Wno-option-ignored -I"$R/include" -I"$R/imagestitch_amd/csrc" \
    harness.cpp "$R/imagestitch_amd/csrc/jpegdec.cpp" "$R/imagestitch_amd/csrc/isx_core.cpp" -o harness
ASAN_OPTIONS=detect_leaks=0 ./harness c/*.jpg
