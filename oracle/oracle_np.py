"""Independent NumPy restatement of the OpenCV-side arithmetic.

TEST INFRASTRUCTURE ONLY.  Written separately from oracle/oracle.c (vectorised, different
code shape) so that the two can be cross-checked bit-exactly in tests/: remap, pyrDown/pyrUp
and MultiBandBlender have no reference golden vectors (OpenCV 3.4.2 is absent: "parity
unpinned"), so agreement of two independent restatements + known-answer tests is the pin.
All float math is done in np.float32 one operation at a time (NumPy never fuses a*b+c).
"""
import numpy as np

F = np.float32


def border_interpolate(p, n, border):
    """cv::borderInterpolate on an int array. border: 0 const(-1) 1 replicate 2 reflect 4 reflect101."""
    p = np.asarray(p, np.int64).copy()
    if border == 1:
        return np.clip(p, 0, n - 1)
    if border in (2, 4):
        if n == 1:
            return np.zeros_like(p)
        d = 1 if border == 4 else 0
        while True:
            neg = p < 0
            big = p >= n
            if not (neg.any() or big.any()):
                break
            p = np.where(neg, -p - 1 + d, np.where(big, n - 1 - (p - n) - d, p))
        return p
    if border == 3:
        return np.mod(p, n)
    return np.where((p >= 0) & (p < n), p, -1)


def cvround(v):
    """cvRound on float32 arrays: round-half-even, NaN / out of range -> INT_MIN."""
    v = np.asarray(v, F)
    ok = np.abs(v) < F(2147483648.0)
    r = np.rint(np.where(ok, v, F(0))).astype(np.int64)
    return np.where(ok, r, -(2 ** 31))


def f2i_trunc(v):
    v = np.asarray(v, F)
    ok = np.abs(v) < F(2147483648.0)
    r = np.trunc(np.where(ok, v, F(0))).astype(np.int64)
    return np.where(ok, r, -(2 ** 31))


def f2s_trunc(v):
    """static_cast<short>(float) as x86 runs it: cvttss2si, keep the low 16 bits."""
    r = f2i_trunc(v) & 0xFFFF
    return np.where(r >= 0x8000, r - 0x10000, r).astype(np.int16)


# ----------------------------------------------------------------------------- remap (A8)
def remap(src, xmap, ymap, interp, border):
    src = np.asarray(src)
    is_u8 = src.dtype == np.uint8
    s3 = src if src.ndim == 3 else src[:, :, None]
    sh, sw, cn = s3.shape
    xm, ym = np.asarray(xmap, F), np.asarray(ymap, F)

    def sat_short(a):
        return np.clip(a, -32768, 32767)

    def fetch(yy, xx):
        ok = (yy >= 0) & (xx >= 0)
        v = s3[np.where(ok, yy, 0), np.where(ok, xx, 0)]
        return np.where(ok[..., None], v, 0)

    if interp == 0:
        sx, sy = sat_short(cvround(xm)), sat_short(cvround(ym))
        inside = (sx >= 0) & (sx < sw) & (sy >= 0) & (sy < sh)
        if border == 0:
            out = fetch(np.where(inside, sy, -1), np.where(inside, sx, -1))
        else:
            out = fetch(border_interpolate(sy, sh, border), border_interpolate(sx, sw, border))
        out = out.astype(src.dtype)
        return out if src.ndim == 3 else out[:, :, 0]

    isx, isy = cvround(xm * F(32)), cvround(ym * F(32))
    fx, fy = isx & 31, isy & 31
    sx, sy = sat_short(isx >> 5), sat_short(isy >> 5)
    if border == 1:
        sx0, sx1 = np.clip(sx, 0, sw - 1), np.clip(sx + 1, 0, sw - 1)
        sy0, sy1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    else:
        sx0, sx1 = border_interpolate(sx, sw, border), border_interpolate(sx + 1, sw, border)
        sy0, sy1 = border_interpolate(sy, sh, border), border_interpolate(sy + 1, sh, border)
    v00, v01, v10, v11 = fetch(sy0, sx0), fetch(sy0, sx1), fetch(sy1, sx0), fetch(sy1, sx1)
    if is_u8:
        # integer table: (32-fx)(32-fy)*32 ..., entry (0,0) = {32767,0,0,1}
        w00 = (32 - fx) * (32 - fy) * 32
        w01 = fx * (32 - fy) * 32
        w10 = (32 - fx) * fy * 32
        w11 = fx * fy * 32
        zero = (fx == 0) & (fy == 0)
        w00 = np.where(zero, 32767, w00)
        w11 = np.where(zero, 1, w11)
        acc = (v00.astype(np.int64) * w00[..., None] + v01.astype(np.int64) * w01[..., None]
               + v10.astype(np.int64) * w10[..., None] + v11.astype(np.int64) * w11[..., None])
        out = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    else:
        ax1 = fx.astype(F) * F(1 / 32)
        ax0 = F(1) - ax1
        ay1 = fy.astype(F) * F(1 / 32)
        ay0 = F(1) - ay1
        w = [(ay0 * ax0)[..., None], (ay0 * ax1)[..., None], (ay1 * ax0)[..., None], (ay1 * ax1)[..., None]]
        out = ((v00.astype(F) * w[0] + v01.astype(F) * w[1]) + v10.astype(F) * w[2]) + v11.astype(F) * w[3]
        out = out.astype(F)
    if border == 0:
        outside = (sx >= sw) | (sx + 1 < 0) | (sy >= sh) | (sy + 1 < 0)
        out = np.where(outside[..., None], 0, out).astype(src.dtype)
    return out if src.ndim == 3 else out[:, :, 0]


# ----------------------------------------------------------------------------- pyramids (A10)
def _wt(a):
    return a.astype(np.int64) if a.dtype == np.int16 else a.astype(F)


def pyr_down(a):
    a = np.asarray(a)
    is_i = a.dtype == np.int16
    s = _wt(a)
    sh, sw = s.shape[:2]
    dh, dw = (sh + 1) // 2, (sw + 1) // 2
    xs = np.arange(dw) * 2
    ix = [border_interpolate(xs + d, sw, 4) for d in (-2, -1, 0, 1, 2)]
    six, four = (6, 4) if is_i else (F(6), F(4))
    h = s[:, ix[2]] * six + (s[:, ix[1]] + s[:, ix[3]]) * four + s[:, ix[0]] + s[:, ix[4]]
    ys = np.arange(dh) * 2
    iy = [border_interpolate(ys + d, sh, 4) for d in (-2, -1, 0, 1, 2)]
    v = h[iy[2]] * six + (h[iy[1]] + h[iy[3]]) * four + h[iy[0]] + h[iy[4]]
    if is_i:
        return np.clip((v + 128) >> 8, -32768, 32767).astype(np.int16)
    return (v * F(1 / 256)).astype(F)


def pyr_up(a):
    a = np.asarray(a)
    is_i = a.dtype == np.int16
    s = _wt(a)
    sh, sw = s.shape[:2]
    c = (lambda k: k) if is_i else F
    row = np.empty((sh, sw * 2) + s.shape[2:], s.dtype)
    if sw == 1:
        row[:, 0] = s[:, 0] * c(8)
        row[:, 1] = s[:, 0] * c(8)
    else:
        row[:, 0] = s[:, 0] * c(6) + s[:, 1] * c(2)
        row[:, 1] = (s[:, 0] + s[:, 1]) * c(4)
        row[:, 2 * sw - 2] = s[:, sw - 2] + s[:, sw - 1] * c(7)
        row[:, 2 * sw - 1] = s[:, sw - 1] * c(8)
        if sw > 2:
            row[:, 2:2 * sw - 2:2] = (s[:, 0:sw - 2] + s[:, 1:sw - 1] * c(6)) + s[:, 2:sw]
            row[:, 3:2 * sw - 2:2] = (s[:, 1:sw - 1] + s[:, 2:sw]) * c(4)
    ys = np.arange(sh)
    up = border_interpolate(2 * (ys - 1), 2 * sh, 4) // 2
    dn = border_interpolate(2 * (ys + 1), 2 * sh, 4) // 2
    r0, r1, r2 = row[up], row, row[dn]
    d0 = (r0 + r1 * c(6)) + r2
    d1 = (r1 + r2) * c(4)
    out = np.empty((sh * 2, sw * 2) + s.shape[2:], s.dtype)
    out[0::2] = d0
    out[1::2] = d1
    if is_i:
        return np.clip((out + 32) >> 6, -32768, 32767).astype(np.int16)
    return (out * F(1 / 64)).astype(F)


def f16_round(a):
    return np.asarray(a, F).astype(np.float16).astype(F)


# ----------------------------------------------------------------------------- MultiBandBlender
class MultiBand:
    I16, F32, F16ACC32 = 0, 1, 2
    EPS = F(1e-5)

    def __init__(self, num_bands=5, precision=0):
        self.actual = num_bands
        self.prec = precision

    def prepare(self, corners, sizes):
        c = np.asarray(corners).reshape(-1, 2)
        s = np.asarray(sizes).reshape(-1, 2)
        tl = c.min(0)
        br = (c + s).max(0)
        w, h = int(br[0] - tl[0]), int(br[1] - tl[1])
        self.fw, self.fh = w, h
        self.L = min(self.actual, int(np.ceil(np.log(float(max(w, h))) / np.log(2.0))))
        m = 1 << self.L
        w += (m - w % m) % m
        h += (m - h % m) % m
        self.roi = (int(tl[0]), int(tl[1]), w, h)
        self.lap, self.wgt = [], []
        rows, cols = h, w
        dt = np.int16 if self.prec == 0 else F
        for _ in range(self.L + 1):
            self.lap.append(np.zeros((rows, cols, 3), dt))
            self.wgt.append(np.zeros((rows, cols), F))
            rows, cols = (rows + 1) // 2, (cols + 1) // 2

    def feed(self, img, mask, tl):
        L, m = self.L, 1 << self.L
        rx, ry, rw, rh = self.roi
        rows, cols = img.shape[:2]
        gap = 3 * m
        tlx, tly = max(rx, tl[0] - gap), max(ry, tl[1] - gap)
        brx, bry = min(rx + rw, tl[0] + cols + gap), min(ry + rh, tl[1] + rows + gap)
        tlx = rx + (((tlx - rx) >> L) << L)
        tly = ry + (((tly - ry) >> L) << L)
        width, height = brx - tlx, bry - tly
        width += (m - width % m) % m
        height += (m - height % m) % m
        brx, bry = tlx + width, tly + height
        dy, dx = max(bry - (ry + rh), 0), max(brx - (rx + rw), 0)
        tlx -= dx; brx -= dx; tly -= dy; bry -= dy
        top, left = tl[1] - tly, tl[0] - tlx
        yy = np.arange(height) - top
        xx = np.arange(width) - left
        sy, sx = border_interpolate(yy, rows, 2), border_interpolate(xx, cols, 2)
        g0 = np.asarray(img)[sy][:, sx]
        inside = ((yy >= 0) & (yy < rows))[:, None] & ((xx >= 0) & (xx < cols))[None, :]
        w0 = np.where(inside, np.asarray(mask)[np.clip(yy, 0, rows - 1)][:, np.clip(xx, 0, cols - 1)].astype(F) * F(1. / 255.), F(0)).astype(F)
        if self.prec == 0:
            g = [g0.astype(np.int16)]
        else:
            g = [g0.astype(F)]
        w = [w0]
        for _ in range(L):
            gn, wn = pyr_down(g[-1]), pyr_down(w[-1])
            if self.prec == 2:
                gn, wn = f16_round(gn), f16_round(wn)
            g.append(gn); w.append(wn)
        for i in range(L):
            up = pyr_up(g[i + 1])
            if self.prec == 0:
                g[i] = np.clip(g[i].astype(np.int64) - up.astype(np.int64), -32768, 32767).astype(np.int16)
            else:
                g[i] = (g[i] - up).astype(F)
        y_tl, y_br, x_tl, x_br = tly - ry, bry - ry, tlx - rx, brx - rx
        for i in range(L + 1):
            rch, rcw = y_br - y_tl, x_br - x_tl
            src = g[i][:rch, :rcw]
            ww = w[i][:rch, :rcw]
            dl = self.lap[i][y_tl:y_tl + rch, x_tl:x_tl + rcw]
            dw = self.wgt[i][y_tl:y_tl + rch, x_tl:x_tl + rcw]
            if self.prec == 0:
                add = f2s_trunc(src.astype(F) * ww[..., None])
                dl[...] = (dl.astype(np.int64) + add.astype(np.int64)).astype(np.int16)  # wraps
            else:
                dl[...] = dl + src * ww[..., None]
            dw[...] = dw + ww
            x_tl //= 2; y_tl //= 2; x_br //= 2; y_br //= 2

    def blend(self, out_f32=False):
        L = self.L
        for i in range(L + 1):
            d = (self.wgt[i] + self.EPS)[..., None]
            if self.prec == 0:
                self.lap[i] = f2s_trunc(self.lap[i].astype(F) / d)
            else:
                self.lap[i] = (self.lap[i] / d).astype(F)
        for i in range(L, 0, -1):
            up = pyr_up(self.lap[i])
            if self.prec == 0:
                self.lap[i - 1] = np.clip(up.astype(np.int64) + self.lap[i - 1].astype(np.int64), -32768, 32767).astype(np.int16)
            else:
                self.lap[i - 1] = (up + self.lap[i - 1]).astype(F)
        on = self.wgt[0][:self.fh, :self.fw] > self.EPS
        out = np.where(on[..., None], self.lap[0][:self.fh, :self.fw], 0)
        mask = np.where(on, 255, 0).astype(np.uint8)
        if self.prec == 0:
            out = out.astype(F) if out_f32 else out.astype(np.int16)
        elif not out_f32:
            out = np.clip(cvround(out.astype(F)), -32768, 32767).astype(np.int16)
        else:
            out = out.astype(F)
        return out, mask
