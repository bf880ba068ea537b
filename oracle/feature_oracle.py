"""CPU oracle of the feature stage of FeatureRegistrator: FAST-9/16 corners with non-maximum
suppression and DAISY descriptors, stated directly from their definitions with numpy.

TEST INFRASTRUCTURE, not product code: only tests/ may import this module.  It shares no code with the product's own
host statement of the stage (microaligner_amd/feature_reg/sparse_cpu.py) nor with the HIP kernels (csrc/daisy.hip):
the tests compare both of them with THIS file.

What the reference calls (microaligner/feature_reg/feature_detection.py:88-118):
    cv.FastFeatureDetector_create(threshold=1, nonmaxSuppression=True, type=TYPE_9_16).detect(tile interior)
    cv.xfeatures2d.DAISY_create(radius=21, q_radius=3, q_theta=8, q_hist=8, norm=NRM_NONE, interpolation=True,
                                use_orientation=False).compute(tile, keypoints)
opencv-contrib 4.5.5.64 (environment.yaml:75) is an un-vendored dependency absent from /root/reference and from this
image.  What pins this oracle instead (tests/golden/make_feature_skimage_golden.py, run under the conda interpreter
that ships scikit-image 0.18.3, fixtures committed):

  * the FAST segment test -- "a pixel is a corner iff 9 contiguous pixels of the 16-pixel Bresenham circle are all
    brighter than centre + t or all darker than centre - t" (Rosten & Drummond 2006; OpenCV's FAST_t<16>) -- is
    compared, as a set of pixels, with skimage.feature.corner_fast(n=9): exact.
  * the machinery DAISY shares with every implementation of Tola, Lepetit & Fua (PAMI 2010) -- Gaussian smoothing of
    8 orientation layers at sigma_r = radius (r + 1) / (2 rings), the ring geometry (radius (r + 1) / rings at angles
    2 pi j / histograms, offsets (rad sin, rad cos)), the layout of the 200 values (centre histogram first, then ring
    by ring, location by location) -- is exercised by daisy_skimage(), which runs scikit-image's variant of the
    descriptor (von-Mises weighted histograms, integer sampling) on the helpers of this file and reproduces
    skimage.feature.daisy to 1e-12.
  * NOT pinned (no implementation of it exists in this image): OpenCV's choice of orientation layers
    (max(cos gx + sin gy, 0) on half central differences), its incremental smoothing schedule and its bilinear
    sampling -- daisy_describe() below states them as the product does; and OpenCV's corner score (largest threshold
    that keeps the pixel a corner) with its strict 3x3 non-maximum suppression, stated here by brute force over
    thresholds.
"""
import numpy as np

# Bresenham circle of radius 3, 16 pixels, clockwise from the pixel below the centre: (dx, dy)
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---- FAST ------------------------------------------------------------------------------------------------
def fast_is_corner(img, t, n=9):
    """Boolean map of the segment test at threshold t (strict inequalities); the 3-pixel border is never a corner."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h, w = img.shape
    out = np.zeros((h, w), bool)
    if h < 7 or w < 7:
        return out
    c = img[3:h - 3, 3:w - 3].astype(np.int32)
    ring = np.stack([img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING])
    brighter, darker = ring > c + t, ring < c - t
    hit = np.zeros(c.shape, bool)
    for start in range(16):
        idx = [(start + k) % 16 for k in range(n)]
        hit |= np.logical_and.reduce(brighter[idx]) | np.logical_and.reduce(darker[idx])
    out[3:h - 3, 3:w - 3] = hit
    return out


def fast_score(img, threshold=1):
    """OpenCV's corner score, by its definition: the largest t at which the pixel still passes the segment test
    (0 where it does not pass at `threshold`).  Brute force over t -- small images only."""
    img = np.asarray(img)
    score = np.zeros(img.shape, np.int32)
    alive = fast_is_corner(img, threshold)
    t = threshold
    while alive.any():
        score[alive] = t
        t += 1
        alive = fast_is_corner(img, t) if t < 255 else np.zeros_like(alive)
    return score


def fast_nms(score):
    """Keep a score only where it is strictly greater than all 8 neighbours (outside the image counts as 0)."""
    h, w = score.shape
    p = np.zeros((h + 2, w + 2), score.dtype)
    p[1:-1, 1:-1] = score
    keep = score > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                keep &= score > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    return np.where(keep, score, 0)


def fast_detect(tile, margin, threshold=1):
    """What the reference's detector sees: the tile without its overlap margin (feature_detection.py:105); returns the
    non-maximum-suppressed score map of that interior."""
    inner = tile[margin:tile.shape[0] - margin, margin:tile.shape[1] - margin]
    return fast_nms(fast_score(inner, threshold))


# ---- Gaussian smoothing as scipy.ndimage.gaussian_filter evaluates it ------------------------------------------------
def gaussian_kernel(sigma, truncate):
    """scipy.ndimage's 1-D Gaussian: radius int(truncate * sigma + 0.5), exp(-x^2 / (2 sigma^2)) normalised to sum 1."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _index(p, n, mode):
    if mode == "nearest":
        return np.clip(p, 0, n - 1)
    assert mode == "reflect"           # half-sample symmetric: d c b a | a b c d | d c b a
    p = np.mod(p, 2 * n)
    return np.where(p >= n, 2 * n - 1 - p, p)


def correlate1d_sym(a, k, axis, mode):
    """Symmetric 1-D correlation along `axis`, float64 accumulation in scipy's order: centre tap first, then tap pairs
    from the outermost inwards, each as (left + right) * weight."""
    a = np.moveaxis(np.asarray(a), axis, -1)
    n = a.shape[-1]
    r = len(k) // 2
    pos = np.arange(n)
    acc = a.astype(np.float64) * k[r]
    for j in range(r, 0, -1):
        left = np.take(a, _index(pos - j, n, mode), axis=-1).astype(np.float64)
        right = np.take(a, _index(pos + j, n, mode), axis=-1).astype(np.float64)
        acc = acc + (left + right) * k[r - j]
    return np.moveaxis(acc, -1, axis)


def gaussian_smooth(planes, sigma, mode, truncate, dtype):
    """gaussian_filter over the last two axes (rows first, then columns), every pass rounded to `dtype`."""
    k = gaussian_kernel(sigma, truncate)
    rows = correlate1d_sym(planes, k, -2, mode).astype(dtype)
    return correlate1d_sym(rows, k, -1, mode).astype(dtype)


# ---- DAISY -----------------------------------------------------------------------------------------------
def ring_offsets(radius, rings, histograms):
    """(dy, dx) of the 1 + rings * histograms histogram locations: centre, then ring by ring."""
    offs = [(0.0, 0.0)]
    for r in range(rings):
        rad = radius * (r + 1) / rings
        for j in range(histograms):
            ang = 2.0 * np.pi * j / histograms
            offs.append((rad * np.sin(ang), rad * np.cos(ang)))
    return np.array(offs)


def ring_sigmas(radius, rings):
    return [radius * (r + 1) / (2.0 * rings) for r in range(rings)]


def daisy_skimage(image, step, radius=21, rings=3, histograms=8, orientations=8):
    """skimage.feature.daisy(image, step, radius, rings, histograms, orientations, normalization='off') of
    scikit-image 0.18.3 on the helpers of this module: forward differences, histograms weighted by the circular normal
    distribution and the gradient magnitude, every histogram smoothed once from the unsmoothed layers (reflect,
    truncate 4), samples at offsets rounded to integers."""
    img = np.asarray(image, np.float64) / (255.0 if np.asarray(image).dtype == np.uint8 else 1.0)
    dx, dy = np.zeros(img.shape), np.zeros(img.shape)
    dx[:, :-1] = np.diff(img, axis=1)
    dy[:-1, :] = np.diff(img, axis=0)
    mag, ori = np.sqrt(dx ** 2 + dy ** 2), np.arctan2(dy, dx)
    kappa = orientations / np.pi
    hist = np.stack([np.exp(kappa * np.cos(ori - (2 * o * np.pi / orientations - np.pi))) * mag for o in range(orientations)])
    sig = ring_sigmas(radius, rings)
    cubes = [gaussian_smooth(hist, s, "reflect", 4.0, np.float64) for s in [sig[0]] + sig]
    offs = ring_offsets(radius, rings, histograms)
    h, w = img.shape
    ys, xs = np.arange(radius, h - radius, step), np.arange(radius, w - radius, step)
    out = np.empty((len(ys), len(xs), len(offs) * orientations))
    for loc, (oy, ox) in enumerate(offs):
        cube = cubes[0] if loc == 0 else cubes[1 + (loc - 1) // histograms]
        yy, xx = ys + int(round(oy)), xs + int(round(ox))
        out[:, :, loc * orientations:(loc + 1) * orientations] = np.moveaxis(cube[:, yy][:, :, xx], 0, -1)
    return out


def daisy_cubes(tile, radius=21, rings=3, orientations=8):
    """The three smoothed orientation cubes of the OpenCV-parameterised descriptor as this repository states it:
    f = tile / 255 in float32; gradients by half central differences (one-sided at the border) in float32; layer o =
    float32(max(cos(th_o) gx + sin(th_o) gy, 0)) with the products in float64; cube r = cube r-1 smoothed by the sigma
    that takes sigma_{r-1} to sigma_r (nearest, truncate 3), every pass rounded to float32."""
    tile = np.asarray(tile)
    f = tile.astype(np.float32) / np.float32(255.0) if tile.dtype == np.uint8 else tile.astype(np.float32)
    h, w = f.shape
    gx, gy = np.empty_like(f), np.empty_like(f)
    gx[:, 1:-1] = (f[:, 2:] - f[:, :-2]) / np.float32(2)
    gx[:, 0], gx[:, -1] = f[:, 1] - f[:, 0], f[:, -1] - f[:, -2]
    gy[1:-1, :] = (f[2:, :] - f[:-2, :]) / np.float32(2)
    gy[0, :], gy[-1, :] = f[1, :] - f[0, :], f[-1, :] - f[-2, :]
    layers = np.empty((orientations, h, w), np.float32)
    for o in range(orientations):
        th = 2.0 * np.pi * o / orientations
        v = np.cos(th) * gx.astype(np.float64) + np.sin(th) * gy.astype(np.float64)
        layers[o] = np.where(v >= 0.0, v, 0.0).astype(np.float32)
    cubes, prev, cur = [], 0.0, layers
    for s in ring_sigmas(radius, rings):
        inc = float(np.sqrt(s * s - prev * prev))
        cur = gaussian_smooth(cur, inc, "nearest", 3.0, np.float32)
        cubes.append(cur)
        prev = s
    return cubes


def daisy_describe(tile, pts_xy, radius=21, rings=3, histograms=8, orientations=8):
    """(n, 200) float32 descriptors at the float (x, y) points: bilinear samples (weights and blend in float32, left to
    right) of cube 0 at the point and of cube r on ring r, coordinates clamped into the tile."""
    cubes = daisy_cubes(tile, radius, rings, orientations)
    P_h, P_w = cubes[0].shape[1:]
    pts = np.asarray(pts_xy, np.float64).reshape(-1, 2)
    offs = ring_offsets(radius, rings, histograms)
    out = np.empty((len(pts), len(offs) * orientations), np.float32)
    for loc, (oy, ox) in enumerate(offs):
        cube = cubes[0] if loc == 0 else cubes[(loc - 1) // histograms]
        ys = np.clip(pts[:, 1] + oy, 0.0, P_h - 1.0)
        xs = np.clip(pts[:, 0] + ox, 0.0, P_w - 1.0)
        y0 = np.minimum(np.floor(ys).astype(np.int64), max(P_h - 2, 0))
        x0 = np.minimum(np.floor(xs).astype(np.int64), max(P_w - 2, 0))
        fy, fx = (ys - y0).astype(np.float32), (xs - x0).astype(np.float32)
        y1, x1 = np.minimum(y0 + 1, P_h - 1), np.minimum(x0 + 1, P_w - 1)
        one = np.float32(1)
        w00, w01, w10, w11 = (one - fy) * (one - fx), (one - fy) * fx, fy * (one - fx), fy * fx
        v = cube[:, y0, x0] * w00 + cube[:, y0, x1] * w01 + cube[:, y1, x0] * w10 + cube[:, y1, x1] * w11
        out[:, loc * orientations:(loc + 1) * orientations] = v.T
    return out
