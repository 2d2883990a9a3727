#!/usr/bin/env python3
"""If THIS box holds an OpenCV (`import cv2`), compare the oracle's restatement of the OpenCV-side arithmetic with OpenCV itself.

The oracle's blend half (SURVEY §8 A8 tie rule, A9 - A12: cv::remap, pyrDown / pyrUp, detail::MultiBandBlender) restates OpenCV 3.4.2 from its
published algorithm; neither the build container nor any GPU box seen so far holds an OpenCV (profiles/round3_opencv_probe.txt), so that half is
"parity unpinned" (DESIGN.md §4).  __graft_entry__.smoke() calls run() on every box it lands on: where `cv2` imports, the comparison below runs
unasked and its verdict is printed with the smoke line; where it does not, one line says so.  Never raises: this is a probe, not a gate.

    python tools/opencv_ab.py            (prints a JSON summary)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def probe():
    """-> (cv2 module or None, one-line description)"""
    try:
        import cv2
        return cv2, "cv2 %s at %s" % (getattr(cv2, "__version__", "?"), getattr(cv2, "__file__", "?"))
    except Exception as e:      # ImportError, or a broken install
        libs = ""
        try:
            import subprocess
            out = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=10).stdout
            libs = ", ".join(sorted({l.split()[0] for l in out.splitlines() if "opencv" in l.lower()})[:4])
        except Exception:
            pass
        return None, "no cv2 (%s)%s" % (type(e).__name__, "; shared libraries: " + libs if libs else "; no libopencv_* in ldconfig")


def run(verbose=True):
    cv2, what = probe()
    res = {"opencv": what, "checks": {}}
    if cv2 is None:
        if verbose:
            print("opencv A/B: " + what + " - the oracle's OpenCV-side arithmetic stays unpinned on this box")
        return res
    import numpy as np
    try:
        from oracle import capi as O
        O.lib()
    except Exception as e:
        res["error"] = "oracle not loadable: %r" % (e,)
        return res
    rng = np.random.default_rng(2026)

    def check(name, fn):
        try:
            res["checks"][name] = fn()
        except Exception as e:      # an API that this OpenCV build lacks, a shape the binding rejects ...
            res["checks"][name] = "error: %s: %s" % (type(e).__name__, str(e)[:160])

    def same(a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape:
            return "shape %s vs %s" % (a.shape, b.shape)
        n = int(np.count_nonzero(a != b))
        return "identical" if n == 0 else "%d of %d values differ (max |d| %s)" % (n, a.size, np.abs(a.astype(np.float64) - b.astype(np.float64)).max())

    # A10: pyrDown / pyrUp, CV_16SC3 and CV_32FC3, odd and even sizes
    for dt, nm in ((np.int16, "s16"), (np.float32, "f32")):
        for (h, w) in ((64, 96), (37, 51)):
            img = (rng.integers(-3000, 3000, (h, w, 3)).astype(dt) if dt == np.int16 else (rng.random((h, w, 3)) * 255).astype(np.float32))
            check("pyrDown_%s_%dx%d" % (nm, w, h), lambda img=img: same(cv2.pyrDown(img), O.pyr_down(img)))
            if h % 2 == 0 and w % 2 == 0:
                small = O.pyr_down(img)
                check("pyrUp_%s_%dx%d" % (nm, w, h), lambda small=small, img=img: same(cv2.pyrUp(small, dstsize=(img.shape[1], img.shape[0])), O.pyr_up(small)))
    # A8: remap LINEAR / BORDER_REFLECT on CV_8UC3 (the CPU path's fixed-point arithmetic and its tie rule), NEAREST / CONSTANT on a mask
    h, w = 120, 160
    src = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    xm = (rng.random((90, 130)) * (w + 20) - 10).astype(np.float32)
    ym = (rng.random((90, 130)) * (h + 20) - 10).astype(np.float32)
    xm[::7, ::5] = np.round(xm[::7, ::5] * 32) / 32 + np.float32(1 / 64)      # coordinates on the 1/64 ties
    check("remap_linear_reflect_u8", lambda: same(cv2.remap(src, xm, ym, cv2.INTER_LINEAR, borderMode=cv2.BORDER_REFLECT), O.remap(src, xm, ym, 1, 2)))
    msk = np.full((h, w), 255, np.uint8)
    check("remap_nearest_constant_mask", lambda: same(cv2.remap(msk, xm, ym, cv2.INTER_NEAREST, borderMode=cv2.BORDER_CONSTANT, borderValue=0), O.remap(msk, xm, ym, 0, 0)))

    # A9 - A12: detail::MultiBandBlender, CV_16SC3 tiles, 5 bands (W:271-273, 281, 302, 313)
    def multiband():
        corners, sizes = [(-40, 7), (233, -12)], [(411, 300), (397, 290)]
        tiles = []
        for (tw, th) in sizes:
            img = rng.integers(0, 256, (th, tw, 3)).astype(np.int16)
            m = np.zeros((th, tw), np.uint8)
            m[:, : tw * 2 // 3] = 255
            tiles.append((img, m))
        ob = O.MultiBand(5, 0)
        ob.prepare(corners, sizes)
        for (img, m), c in zip(tiles, corners):
            ob.feed(img, m, c)
        od, om = ob.blend(False)
        mk = getattr(cv2, "detail_MultiBandBlender", None) or getattr(cv2.detail, "MultiBandBlender")
        b = mk(0, 5)
        b.prepare(cv2.detail.resultRoi(corners=corners, sizes=sizes))
        for (img, m), c in zip(tiles, corners):
            b.feed(img, m, c)
        d, dm = b.blend(None, None)
        return {"image": same(d, od), "mask": same(dm, om)}
    check("multiband_s16_5bands", multiband)
    if verbose:
        print("opencv A/B (" + what + "): " + json.dumps(res["checks"]))
    return res


if __name__ == "__main__":
    print(json.dumps(run(verbose=False), indent=1))
