"""Synthetic inputs for the DP seam estimate (estimateSeam S:806-957): two overlapping tiles, a union-sized label image
with one intersection component (ragged border, holes) and two seam tips on it."""
import numpy as np


def make_case(seed, size1=(90, 120), size2=(100, 110), tl1=(-30, 5), tl2=(40, -8), u8=False, horizontal=False, swap=False, holes=True, label=3):
    rng = np.random.default_rng(seed)
    (h1, w1), (h2, w2) = size1, size2
    utl = (min(tl1[0], tl2[0]), min(tl1[1], tl2[1]))
    ubr = (max(tl1[0] + w1, tl2[0] + w2), max(tl1[1] + h1, tl2[1] + h2))
    uw, uh = ubr[0] - utl[0], ubr[1] - utl[1]
    ix0, iy0 = max(tl1[0], tl2[0]) - utl[0], max(tl1[1], tl2[1]) - utl[1]
    ix1, iy1 = min(tl1[0] + w1, tl2[0] + w2) - utl[0], min(tl1[1] + h1, tl2[1] + h2) - utl[1]
    assert ix1 - ix0 >= 4 and iy1 - iy0 >= 4
    labels = np.zeros((uh, uw), np.int32)
    labels[: , :] = 1
    labels[iy0:iy1, ix0:ix1] = label
    if holes:   # ragged border and a few islands of another component inside the intersection
        for _ in range(6):
            y, x = int(rng.integers(iy0, iy1)), int(rng.integers(ix0, ix1))
            hh, ww = int(rng.integers(1, max(2, (iy1 - iy0) // 6))), int(rng.integers(1, max(2, (ix1 - ix0) // 6)))
            labels[y:min(y + hh, iy1), x:min(x + ww, ix1)] = 7
        labels[iy0:iy0 + 2, ix0:ix0 + (ix1 - ix0) // 3] = 1
    ys, xs = np.nonzero(labels == label)
    rx, ry = int(xs.min()), int(ys.min())
    roi = (rx, ry, int(xs.max()) + 1 - rx, int(ys.max()) + 1 - ry)      # Rect(tls_[comp], brs_[comp]): brs exclusive
    if horizontal:   # tips on the left / right side
        ca, cb = xs.min(), xs.max()
        ya, yb = ys[xs == ca], ys[xs == cb]
        p1, p2 = (int(ca), int(ya[len(ya) // 2])), (int(cb), int(yb[len(yb) // 3]))
    else:            # tips on the top / bottom side
        ra, rb = ys.min(), ys.max()
        xa, xb = xs[ys == ra], xs[ys == rb]
        p1, p2 = (int(xa[len(xa) // 2]), int(ra)), (int(xb[len(xb) // 3]), int(rb))
    if swap:
        p1, p2 = p2, p1
    if u8:
        img1 = rng.integers(0, 256, (h1, w1, 3)).astype(np.uint8)
        img2 = rng.integers(0, 256, (h2, w2, 3)).astype(np.uint8)
    else:   # smooth content + noise, as warped photographs converted to CV_32F (W:261)
        yy, xx = np.mgrid[0:h1, 0:w1]
        img1 = (128 + 60 * np.sin(xx / 9.0)[..., None] * np.cos(yy / 7.0)[..., None] + rng.normal(0, 8, (h1, w1, 3))).astype(np.float32)
        yy, xx = np.mgrid[0:h2, 0:w2]
        img2 = (128 + 60 * np.sin((xx + tl2[0] - tl1[0]) / 9.0)[..., None] * np.cos((yy + tl2[1] - tl1[1]) / 7.0)[..., None] + rng.normal(0, 8, (h2, w2, 3))).astype(np.float32)
    return dict(img1=img1, img2=img2, tl1=tl1, tl2=tl2, union_tl=utl, labels=labels, label=label, roi=roi, p1=p1, p2=p2)


def make_find_case(seed, n_images=2, u8=False, holes=True, size=(110, 140)):
    """Inputs of DpSeamFinder::find (S:87-124): n overlapping tiles in a row with barrel-shaped masks (as a cylindrical warp
    leaves them), optional holes (extra components), smooth + noisy CV_32FC3 (or CV_8UC3) content."""
    rng = np.random.default_rng(seed)
    h0, w0 = size
    images, masks, corners = [], [], []
    x = 0
    for i in range(n_images):
        h, w = h0 + int(rng.integers(-8, 9)), w0 + int(rng.integers(-10, 11))
        tl = (x + int(rng.integers(-3, 4)), int(rng.integers(-6, 7)))
        x += int(w * rng.uniform(0.45, 0.7))
        yy, xx = np.mgrid[0:h, 0:w]
        bulge = 0.00035 * rng.uniform(0.5, 1.5) * (xx - w / 2.0) ** 2
        m = ((yy >= bulge) & (yy <= h - 1 - bulge)).astype(np.uint8) * 255
        if holes:
            for _ in range(int(rng.integers(0, 4))):
                cy, cx = int(rng.integers(0, h)), int(rng.integers(0, w))
                m[cy:cy + int(rng.integers(1, 9)), cx:cx + int(rng.integers(1, 12))] = 0
        gx, gy = xx + tl[0], yy + tl[1]
        base = 128 + 55 * np.sin(gx / 11.0)[..., None] * np.cos(gy / 8.0)[..., None] + np.stack([gx * 0.1, gy * 0.2, (gx + gy) * 0.05], -1)
        img = base + rng.normal(0, 9 + 3 * i, (h, w, 3))
        img = np.clip(img, 0, 255)
        images.append(np.rint(img).astype(np.uint8) if u8 else np.rint(img).astype(np.float32))
        masks.append(m)
        corners.append(tl)
    return images, corners, masks
