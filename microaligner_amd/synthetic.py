"""Seeded synthetic image pairs: Gaussian-smoothed noise texture, reference =
canvas crop, moving = canvas resampled at p + d(p) with d = global shift + smooth field, so that the
true flow (mov(p) ~ ref(p + flow(p))) is d.  Used by the tests, the golden-vector script and bench.py.
"""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from scipy.ndimage import gaussian_filter1d

GLOBAL_SHIFT = (3.3, -2.1)  # (dx, dy) in pixels


def displacement(H, W, shift=GLOBAL_SHIFT, amp=2.0, dtype=np.float32):
    """d(p) = shift + amp * (sin(2*pi*y/H*3), cos(2*pi*x/W*2)); returns (dx, dy) as (H,1)/(1,W)-broadcastable."""
    y = np.arange(H, dtype=np.float64)[:, None]
    x = np.arange(W, dtype=np.float64)[None, :]
    dx = shift[0] + amp * np.sin(2 * np.pi * y / H * 3) + 0 * x
    dy = shift[1] + amp * np.cos(2 * np.pi * x / W * 2) + 0 * y
    return dx.astype(dtype), dy.astype(dtype)


def _bilinear(canvas, ys, xs):
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0).astype(np.float32)
    fx = (xs - x0).astype(np.float32)
    y0 = np.clip(y0, 0, canvas.shape[0] - 2)
    x0 = np.clip(x0, 0, canvas.shape[1] - 2)
    a = canvas[y0, x0]
    b = canvas[y0, x0 + 1]
    c = canvas[y0 + 1, x0]
    d = canvas[y0 + 1, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def _smooth(canvas, sigma, band=512):
    """scipy.ndimage.gaussian_filter(canvas, sigma, mode="reflect"), bit for bit, but with each 1-D pass cut into
    independent column / row bands that run on a thread pool (scipy releases the GIL)."""
    tmp, out = np.empty_like(canvas), np.empty_like(canvas)
    H, W = canvas.shape

    def cols(x0):
        gaussian_filter1d(canvas[:, x0:x0 + band], sigma, axis=0, mode="reflect", output=tmp[:, x0:x0 + band])

    def rows(y0):
        gaussian_filter1d(tmp[y0:y0 + band], sigma, axis=1, mode="reflect", output=out[y0:y0 + band])

    with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as ex:
        list(ex.map(cols, range(0, W, band)))
        list(ex.map(rows, range(0, H, band)))
    return out


def make_pair(H, W, seed=1, dtype=np.float32, shift=GLOBAL_SHIFT, amp=2.0, margin=20, sigma=4.0, band=512):
    """(ref, mov) of shape (H, W).  float32 in [0, 255]; uint8/uint16 are rounded casts (uint16 x 257)."""
    rng = np.random.default_rng(seed)
    canvas = rng.standard_normal((H + 2 * margin, W + 2 * margin), dtype=np.float32)
    canvas = _smooth(canvas, sigma)
    lo, hi = float(canvas.min()), float(canvas.max())
    canvas = (canvas - lo) * (255.0 / (hi - lo))
    ref = np.ascontiguousarray(canvas[margin:margin + H, margin:margin + W])
    mov = np.empty((H, W), np.float32)

    def resample(y0):  # banded: bounds host memory on 16k x 16k and runs on the thread pool
        y1 = min(y0 + band, H)
        yy = np.arange(y0, y1, dtype=np.float64)[:, None]
        xx = np.arange(W, dtype=np.float64)[None, :]
        dx = shift[0] + amp * np.sin(2 * np.pi * yy / H * 3)
        dy = shift[1] + amp * np.cos(2 * np.pi * xx / W * 2)
        mov[y0:y1] = _bilinear(canvas, yy + margin + dy, xx + margin + dx)

    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(resample, range(0, H, band)))
    return _cast(ref, dtype), _cast(mov, dtype)


def make_unrelated_pair(H, W, seed=1, dtype=np.float32):
    """Two independent textures: registration cannot help, the gate must reject."""
    a, _ = make_pair(H, W, seed, np.float32)
    b, _ = make_pair(H, W, seed + 1000, np.float32)
    return _cast(a, dtype), _cast(b, dtype)


def _cast(img, dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return np.ascontiguousarray(img, dtype=np.float32)
    if dtype == np.uint8:
        return np.clip(np.rint(img), 0, 255).astype(np.uint8)
    if dtype == np.uint16:
        return np.clip(np.rint(img * 257.0), 0, 65535).astype(np.uint16)
    raise ValueError(f"unsupported dtype {dtype}")


def make_cells(H, W, n=None, seed=1, dtype=np.uint16):
    """Fluorescence-like test image: `n` blurred point sources of random brightness (two spot sizes) plus a little
    noise.  Unlike the smooth field of make_pair it has distinctive local constellations, which is what the
    feature-based registration needs."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    n = n if n is not None else max(H * W // 400, 16)
    pts = np.zeros((H, W), np.float32)
    pts[rng.integers(0, H, n), rng.integers(0, W, n)] = rng.uniform(0.3, 1.0, n).astype(np.float32)
    img = gaussian_filter(pts, 2.0) + 0.5 * gaussian_filter(np.roll(pts, 7, 0), 4.5)
    img += rng.normal(0, 0.02 * float(img.max()), img.shape).astype(np.float32)
    img = np.clip(img / float(img.max()), 0, 1)
    if np.dtype(dtype) == np.uint8:
        return (img * 255).astype(np.uint8)
    if np.dtype(dtype) == np.uint16:
        return (img * 60000).astype(np.uint16)
    return (img * 255).astype(np.float32)


def mosaic_tile_transform(H, W, seed):
    """The known similarity a mosaic tile of BASELINE cfg5 is misplaced by: rotation <= 0.5 deg about the tile
    centre, shift <= 20 px.  Returns the 2x3 matrix M with mov(p) = ref(M^-1 p) away from the border, i.e.
    cv2.warpAffine(ref, M) -- the transform FeatureRegistrator.register() is expected to undo (it returns ~M^-1)."""
    rng = np.random.default_rng(1000 + seed)
    ang = np.deg2rad(rng.uniform(-0.5, 0.5))
    tx, ty = rng.uniform(-20, 20, 2)
    c, s = np.cos(ang), np.sin(ang)
    cx, cy = W / 2.0, H / 2.0
    return np.array([[c, -s, tx + cx - c * cx + s * cy], [s, c, ty + cy - s * cx - c * cy]], np.float64)


def make_mosaic_tile(H, W, seed, dtype=np.uint16, amp=1.5):
    """(ref, mov, M) of one mosaic tile of cfg5: cell-like reference, moving image = the reference under the known
    similarity M (bilinear, zero border) followed by a small smooth non-linear displacement of amplitude `amp` px,
    which is what the optical-flow stage is there to remove."""
    from scipy.ndimage import map_coordinates
    ref = make_cells(H, W, seed=seed, dtype=np.float32)
    M = mosaic_tile_transform(H, W, seed)
    inv = np.linalg.inv(np.vstack([M, [0, 0, 1]]))[:2]
    mov = np.empty((H, W), np.float32)
    band = 512

    def rows(y0):
        y1 = min(y0 + band, H)
        yy = np.arange(y0, y1, dtype=np.float64)[:, None]
        xx = np.arange(W, dtype=np.float64)[None, :]
        xs = xx + amp * np.sin(yy / 97.0)               # smooth residual, then the similarity
        ys = yy + amp * np.cos(xx / 113.0)
        sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
        sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
        mov[y0:y1] = map_coordinates(ref, [sy, sx], order=1, mode="constant", cval=0.0, prefilter=False)

    with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(rows, range(0, H, band)))
    scale = {np.dtype(np.uint8): 1.0, np.dtype(np.uint16): 60000 / 255.0, np.dtype(np.float32): 1.0}[np.dtype(dtype)]
    cast = (lambda a: a) if np.dtype(dtype) == np.float32 else (lambda a: np.clip(np.rint(a * scale), 0, np.iinfo(dtype).max).astype(dtype))
    return cast(ref), cast(mov), M
