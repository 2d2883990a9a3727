"""Synthetic workloads of SURVEY.md §8(d): camera pairs, noisy sinusoid tiles, sinusoidal seam masks.

Used by bench.py, __graft_entry__.smoke() and the tests.  Pure numpy; no reference data involved.
"""
import numpy as np

SEED0 = 0xC0FFEE


def _rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float64)


def camera_pair(width, height, focal, yaw=0.36, pitch=0.010, roll=0.005):
    """K = [f 0 w/2; 0 f h/2; 0 0 1], R_i = Ry(-/+yaw) Rx(pitch) Rz(roll) as CV_32F (W:183-188,225-226)."""
    K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], np.float32)
    Rs = [(_rot("y", s * yaw) @ _rot("x", pitch) @ _rot("z", roll)).astype(np.float32) for s in (-1.0, 1.0)]
    return K, Rs


def make_tile(height, width, tile_index=0, noise_only=False):
    """u8 = clamp(128 + 64 sin(2 pi x / 257) cos(2 pi y / 193) + noise), noise in U{-32..31} from a
    PCG64 stream seeded SEED0 + tile_index; the three channels are phase shifted.
    noise_only: pure U{0..255} (worst case for rounding parity)."""
    rng = np.random.Generator(np.random.PCG64(SEED0 + tile_index))
    if noise_only:
        return rng.integers(0, 256, (height, width, 3), dtype=np.uint8)
    noise = rng.integers(-32, 32, (height, width, 3), dtype=np.int16)
    y = np.arange(height, dtype=np.float32)[:, None]
    x = np.arange(width, dtype=np.float32)[None, :]
    out = np.empty((height, width, 3), np.uint8)
    for c in range(3):
        base = 128.0 + 64.0 * np.sin(2 * np.pi * x / 257.0 + c * 0.7) * np.cos(2 * np.pi * y / 193.0 + c * 0.4)
        out[:, :, c] = np.clip(np.rint(base).astype(np.int16) + noise[:, :, c], 0, 255).astype(np.uint8)
    return out


def camera_ring(width, height, focal, n, yaw_step, pitch=0.010, roll=0.005):
    """n cameras on one rig, yaw_i = (i - (n - 1) / 2) * yaw_step (BASELINE config 5: n = 8, yaw step 0.55 rad);
    n = 2 with yaw_step = 2 * yaw is camera_pair."""
    K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1]], np.float32)
    Rs = [(_rot("y", (i - (n - 1) / 2.0) * yaw_step) @ _rot("x", pitch) @ _rot("z", roll)).astype(np.float32) for i in range(n)]
    return K, Rs


def seam_masks(corners, warped_masks, sizes=None):
    """Seam-finder stand-in for a row of n >= 2 tiles: between neighbours (in order of their corner x) the
    seam x_s(y) = x_mid + round(40 sin(2 pi y / 512)) runs through the centre of their overlap; tile i keeps
    warped_i & (x_s[i-1] <= X < x_s[i]) (X, y in panorama coordinates).  For a pair: mask0 = warped0 & (X < x_s),
    mask1 = warped1 & (X >= x_s).  `sizes` ((w, h) per tile) lets entries of warped_masks be None: those tiles get no mask (a
    rank that blends one strip of the panorama holds only the tiles near it; the seams still depend on every tile's rectangle)."""
    n = len(corners)
    widths = [warped_masks[i].shape[1] if sizes is None else sizes[i][0] for i in range(n)]
    order = sorted(range(n), key=lambda i: corners[i][0])
    mids = []
    for a, b in zip(order[:-1], order[1:]):
        ov_l = max(corners[a][0], corners[b][0])
        ov_r = min(corners[a][0] + widths[a], corners[b][0] + widths[b])
        mids.append((ov_l + ov_r) // 2)
    out = [None] * n
    for pos, i in enumerate(order):
        (cx, cy), m = corners[i], warped_masks[i]
        if m is None:
            continue
        Y = cy + np.arange(m.shape[0])[:, None]
        X = cx + np.arange(m.shape[1])[None, :]
        wob = np.rint(40.0 * np.sin(2 * np.pi * Y / 512.0)).astype(np.int64)
        keep = np.ones(m.shape, bool)
        if pos > 0:
            keep &= X >= mids[pos - 1] + wob
        if pos < n - 1:
            keep &= X < mids[pos] + wob
        out[i] = np.where(keep, m, 0).astype(np.uint8)
    return out
