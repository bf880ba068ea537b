"""CPU implementation of the sparse half of FeatureRegistrator: FAST-9/16 corners, DAISY
descriptors, 2-nearest-neighbour matching with the ratio test and a RANSAC partial-affine fit.

The reference obtains these from opencv-contrib (feature_reg/feature_detection.py:88-158):
cv.FastFeatureDetector_create(threshold=1, nonmaxSuppression=True, TYPE_9_16), cv.xfeatures2d.DAISY_create(radius=21,
q_radius=3, q_theta=8, q_hist=8, NRM_NONE, interpolation=True, use_orientation=False), cv.FlannBasedMatcher +
Lowe's ratio 0.5, cv.estimateAffinePartial2D(RANSAC, confidence=0.99).  opencv-contrib is not available to this
build, so the stage is restated here from the published algorithms.  This module is the HOST statement of that
stage and the definition its device counterparts are tested against: csrc/daisy.hip (FAST score map + non-maximum
suppression, corner selection, DAISY layers / smoothing / sampling, all tiles of a level in one call), csrc/knn.hip (the
exact 2-NN search) and csrc/ransac.hip (ratio test, RANSAC similarity fit) reproduce it bit for bit and are what
FeatureRegistrator runs; this file serves host arrays (tiles that are not uint8 device images, coordinates that are not
integer-valued) and the tests.  PARITY UNPINNED: FAST follows OpenCV's segment test, score and 3x3 non-maximum suppression exactly
as published; DAISY follows Tola et al. (PAMI 2010) with OpenCV's parameter meaning but not its exact smoothing
schedule; matching is exact 2-NN where FLANN is approximate; RANSAC uses numpy's seeded sequence and closed-form fits where OpenCV has its
own generator and a Levenberg-Marquardt refinement.  The outputs are
therefore functionally equivalent (same kind of keypoints, descriptors and 2x3 similarity transform), not
bit-identical to opencv-contrib.
"""
import threading
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

_BLAS_LOCK = threading.Lock()

# Bresenham circle of radius 3, OpenCV's order (dx, dy)
_RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
         (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


@dataclass
class KeyPoint:
    """The fields of cv2.KeyPoint that the reference carries around (feature_detection.py:38-71)."""
    pt: Tuple[float, float]
    size: float = 7.0
    angle: float = -1.0
    response: float = 0.0
    octave: int = 0
    class_id: int = -1


def fast_score_map(img: np.ndarray, threshold: int = 1) -> np.ndarray:
    """Corner score of every pixel (0 where the 9-of-16 segment test fails), OpenCV's cornerScore<16>:
    the largest t for which the pixel is still a corner, i.e. max over the 16 arcs of 9 contiguous ring pixels of
    min(v - ring) (darker arc) or min(ring - v) (brighter arc), minus 1.  A pixel is a corner iff that maximum
    exceeds `threshold`.  The 3-pixel border is never a corner."""
    img = np.asarray(img)
    if img.dtype != np.uint8:
        raise ValueError("FAST works on uint8 images (the DOG output)")
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    if h < 7 or w < 7:
        return score
    c = img[3:h - 3, 3:w - 3].astype(np.int16)
    d = np.empty((16,) + c.shape, np.int16)
    for k, (dx, dy) in enumerate(_RING):
        d[k] = c - img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx]
    # circular sliding minimum / maximum over 9 consecutive ring positions by doubling (2, 4, 8, then 9)
    def windows(op):
        m2 = [op(d[s], d[(s + 1) % 16]) for s in range(16)]
        m4 = [op(m2[s], m2[(s + 2) % 16]) for s in range(16)]
        m8 = [op(m4[s], m4[(s + 4) % 16]) for s in range(16)]
        return [op(m8[s], d[(s + 8) % 16]) for s in range(16)]
    best = None
    for m in windows(np.minimum):          # darker arcs: all (v - ring) large
        best = m if best is None else np.maximum(best, m)
    for m in windows(np.maximum):          # brighter arcs: all (ring - v) large  ==  -(max of v - ring)
        best = np.maximum(best, -m)
    inner = np.where(best > threshold, best.astype(np.int32) - 1, 0)
    score[3:h - 3, 3:w - 3] = inner
    return score


def fast_detect(img: np.ndarray, threshold: int = 1, nonmax: bool = True) -> List[KeyPoint]:
    """cv.FastFeatureDetector_create(threshold, nonmaxSuppression, TYPE_9_16).detect(img): keypoints in row-major
    order, response = corner score, size 7.  With nonmax a corner survives iff its score is strictly greater than
    the scores of its 8 neighbours (non-corners count as 0)."""
    s = fast_score_map(img, threshold)
    ok = s > 0
    if nonmax:
        p = np.pad(s, 1)
        h, w = s.shape
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx or dy:
                    ok &= s > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ys, xs = np.nonzero(ok)
    return [KeyPoint((float(x), float(y)), 7.0, -1.0, float(s[y, x]), 0, -1) for y, x in zip(ys, xs)]


class Daisy:
    """DAISY descriptor (Tola, Lepetit, Fua 2010) at given points: 1 + q_radius * q_theta histograms of q_hist
    orientation bins = 200 floats with the reference's parameters, no normalisation (NRM_NONE), bilinear sampling
    (interpolation=True), fixed orientation (use_orientation=False)."""

    def __init__(self, radius=21, q_radius=3, q_theta=8, q_hist=8):
        self.radius, self.q_radius, self.q_theta, self.q_hist = radius, q_radius, q_theta, q_hist

    def smoothing_increments(self) -> List[float]:
        """Sigma of the Gaussian that takes cube r-1 to cube r (cube -1 = the orientation layers)."""
        sigmas = [self.radius * (r + 1) / (2.0 * self.q_radius) for r in range(self.q_radius)]
        prev, incs = 0.0, []
        for s in sigmas:
            incs.append(float(np.sqrt(s * s - prev * prev)))
            prev = s
        return incs

    def sample_offsets(self) -> np.ndarray:
        """(dy, dx) of the 1 + q_radius * q_theta histogram locations relative to the keypoint, in compute()'s order."""
        offs = [(0.0, 0.0)]
        for r in range(self.q_radius):
            rad = self.radius * (r + 1) / self.q_radius
            for j in range(self.q_theta):
                ang = 2.0 * np.pi * j / self.q_theta
                offs.append((rad * np.sin(ang), rad * np.cos(ang)))
        return np.array(offs, np.float64)

    def _cubes(self, img: np.ndarray) -> List[np.ndarray]:
        from scipy.ndimage import gaussian_filter
        f = img.astype(np.float32) / (255.0 if img.dtype == np.uint8 else 1.0)
        gy, gx = np.gradient(f)
        layers = np.empty((self.q_hist,) + f.shape, np.float32)
        for o in range(self.q_hist):
            th = 2.0 * np.pi * o / self.q_hist
            np.maximum(np.cos(th) * gx + np.sin(th) * gy, 0, out=layers[o])
        sigmas = [self.radius * (r + 1) / (2.0 * self.q_radius) for r in range(self.q_radius)]
        cubes, prev, cur = [], 0.0, layers
        for s in sigmas:
            inc = float(np.sqrt(s * s - prev * prev))
            cur = gaussian_filter(cur, sigma=(0, inc, inc), mode="nearest", truncate=3.0)
            cubes.append(cur)
            prev = s
        return cubes

    @staticmethod
    def _sample(cube: np.ndarray, ys: np.ndarray, xs: np.ndarray) -> np.ndarray:
        """Bilinear sample of every layer at (ys, xs) (clamped to the image): (n, q_hist)."""
        _, h, w = cube.shape
        ys = np.clip(ys, 0, h - 1.0)
        xs = np.clip(xs, 0, w - 1.0)
        y0 = np.minimum(np.floor(ys).astype(np.int64), h - 2) if h > 1 else np.zeros(ys.shape, np.int64)
        x0 = np.minimum(np.floor(xs).astype(np.int64), w - 2) if w > 1 else np.zeros(xs.shape, np.int64)
        fy = (ys - y0).astype(np.float32)
        fx = (xs - x0).astype(np.float32)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        v = (cube[:, y0, x0] * ((1 - fy) * (1 - fx)) + cube[:, y0, x1] * ((1 - fy) * fx) +
             cube[:, y1, x0] * (fy * (1 - fx)) + cube[:, y1, x1] * (fy * fx))
        return v.T

    def compute(self, img: np.ndarray, keypoints: List[KeyPoint]) -> Optional[np.ndarray]:
        if not keypoints:
            return None
        cubes = self._cubes(img)
        pts = np.array([kp.pt for kp in keypoints], np.float64)
        xs, ys = pts[:, 0], pts[:, 1]
        n = len(keypoints)
        des = np.empty((n, (1 + self.q_radius * self.q_theta) * self.q_hist), np.float32)
        des[:, :self.q_hist] = self._sample(cubes[0], ys, xs)
        col = self.q_hist
        for r in range(self.q_radius):
            rad = self.radius * (r + 1) / self.q_radius
            for j in range(self.q_theta):
                ang = 2.0 * np.pi * j / self.q_theta
                des[:, col:col + self.q_hist] = self._sample(cubes[r], ys + rad * np.sin(ang), xs + rad * np.cos(ang))
                col += self.q_hist
        return des


def knn2(query: np.ndarray, train: np.ndarray, block: int = 1024):
    """Exact two nearest neighbours (L2) of every query row in `train`: (idx (n,2), dist (n,2)).  Needs at least
    two train rows.  Two arg-min passes per block (the first minimum is masked for the second)."""
    q = np.ascontiguousarray(query, np.float32)
    t = np.ascontiguousarray(train, np.float32)
    tn = np.einsum("ij,ij->i", t, t)
    idx = np.empty((len(q), 2), np.int64)
    dist = np.empty((len(q), 2), np.float32)
    # threadpool_limits is process-global and not re-entrant across threads: one search at a time holds the cap
    with _BLAS_LOCK, _blas_threads(16):
        _knn2_blocks(q, t, tn, idx, dist, block)
    return idx, dist


def knn2_sequential(query: np.ndarray, train: np.ndarray, block: int = 256):
    """The DEFINITION the device search (ma_knn2_l2) implements, restated with numpy: squared distance accumulated in
    float32 over ascending dimension, d2 = d2 + diff * diff with the product and the sum rounded separately; the two
    smallest per query, ties to the lower index; distances are the float32 roots.  Slow (one pass over the distance
    block per dimension): for tests and for generating fixtures, not for 45 000 x 45 000 problems."""
    q = np.ascontiguousarray(query, np.float32)
    tt = np.ascontiguousarray(np.asarray(train, np.float32).T)      # (dim, nt): contiguous rows per dimension
    idx = np.empty((len(q), 2), np.int64)
    dist = np.empty((len(q), 2), np.float32)
    for s in range(0, len(q), block):
        qb = q[s:s + block]
        d2 = np.zeros((len(qb), tt.shape[1]), np.float32)
        diff = np.empty_like(d2)
        for k in range(tt.shape[0]):
            np.subtract(qb[:, k, None], tt[k][None, :], out=diff)
            np.multiply(diff, diff, out=diff)
            np.add(d2, diff, out=d2)
        rows = np.arange(len(qb))
        i0 = d2.argmin(1)
        v0 = d2[rows, i0].copy()
        d2[rows, i0] = np.inf
        i1 = d2.argmin(1)          # among equal values argmin returns the lowest index, and i0 < i1 on exact ties
        idx[s:s + block, 0], idx[s:s + block, 1] = i0, i1
        dist[s:s + block, 0], dist[s:s + block, 1] = np.sqrt(v0), np.sqrt(d2[rows, i1])
    return idx, dist


def _blas_threads(n):
    """Caps the BLAS thread pool for the block products (hundreds of spinning threads on a many-core host are
    slower than 16 for these sizes, and pathological when the host is shared); a no-op without threadpoolctl."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n, user_api="blas")
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def _knn2_blocks(q, t, tn, idx, dist, block):
    for s in range(0, len(q), block):
        qb = q[s:s + block]
        d2 = np.einsum("ij,ij->i", qb, qb)[:, None] + tn[None, :] - 2.0 * (qb @ t.T)
        rows = np.arange(len(qb))
        i0 = d2.argmin(1)
        v0 = d2[rows, i0].copy()
        d2[rows, i0] = np.inf
        i1 = d2.argmin(1)
        v1 = d2[rows, i1]
        idx[s:s + block, 0], idx[s:s + block, 1] = i0, i1
        dist[s:s + block, 0] = np.sqrt(np.maximum(v0, 0))
        dist[s:s + block, 1] = np.sqrt(np.maximum(v1, 0))


def _fit_similarity(src: np.ndarray, dst: np.ndarray) -> Optional[np.ndarray]:
    """Least-squares 4-DOF transform dst ~ [[a, -b, tx], [b, a, ty]] @ (src, 1), in closed form.

    The normal equations of  sum |a x - b y + tx - u|^2 + |b x + a y + ty - v|^2  about an integer-valued centre
    (cx, cy, cu, cv) = floor(mean):  a = (n Sxu - (Sx Su + Sy Sv)) / D,  b = (n Sxv - (Sx Sv - Sy Su)) / D,
    D = n Sxx - (Sx^2 + Sy^2)  with the centred sums  Sx = sum x', Sxx = sum (x'^2 + y'^2), Sxu = sum (x' u' + y' v'),
    Sxv = sum (x' v' - y' u').  Keypoints are integer pixel positions: every term and every partial sum is an integer below
    2^53, i.e. EXACT in float64 in any summation order -- which is what lets the device (csrc/ransac.hip, a parallel
    reduction) and this file (numpy's pairwise sums) produce the same bits.  The few operations after the sums are written
    out one by one in the order the kernel uses.  None when the source points coincide (rank < 4)."""
    n = len(src)
    if n < 2:
        return None
    x, y, u, v = src[:, 0], src[:, 1], dst[:, 0], dst[:, 1]
    fn = float(n)
    cx, cy = float(np.floor(x.sum() / fn)), float(np.floor(y.sum() / fn))
    cu, cv = float(np.floor(u.sum() / fn)), float(np.floor(v.sum() / fn))
    x, y, u, v = x - cx, y - cy, u - cu, v - cv
    Sx, Sy, Su, Sv = float(x.sum()), float(y.sum()), float(u.sum()), float(v.sum())
    Sxx = float((x * x + y * y).sum())
    Sxu = float((x * u + y * v).sum())
    Sxv = float((x * v - y * u).sum())
    D = fn * Sxx - (Sx * Sx + Sy * Sy)
    if not D > 0.0:
        return None
    a = (fn * Sxu - (Sx * Su + Sy * Sv)) / D
    b = (fn * Sxv - (Sx * Sv - Sy * Su)) / D
    tx = (Su - (a * Sx - b * Sy)) / fn          # translation between the centred frames ...
    ty = (Sv - (b * Sx + a * Sy)) / fn
    tx = (tx + cu) - (a * cx - b * cy)          # ... and between the original ones
    ty = (ty + cv) - (b * cx + a * cy)
    return np.array([[a, -b, tx], [b, a, ty]], np.float64)


def _similarity_inliers(M, sx, sy, dx, dy, thr2):
    """Points whose squared residual under M is below thr2; the expression the RANSAC loop, the refinement and the device
    kernel share: ex = ((a x - b y) + tx) - u, ey = ((b x + a y) + ty) - v, err = ex ex + ey ey, products and sums rounded
    one by one (no fused multiply-add, no BLAS)."""
    a, b, tx, ty = float(M[0, 0]), float(M[1, 0]), float(M[0, 2]), float(M[1, 2])
    ex, ey = a * sx - b * sy + tx - dx, b * sx + a * sy + ty - dy
    return ex * ex + ey * ey < thr2


def ransac_iterations(count: int, n: int, confidence: float, max_iters: int, it: int) -> int:
    """Iterations RANSAC still needs once a model with `count` inliers of `n` points has been seen (two-point samples):
    ceil(log(1 - confidence) / log(1 - w^2)), w = count / n, at most max_iters; `it` (stop now) when every point is an
    inlier.  math.log, i.e. the C library's: the host half of the device path (csrc/ransac.hip) calls the same function."""
    import math
    w = count / n
    denom = math.log(max(1.0 - w * w, 1e-12))
    return min(max_iters, int(math.ceil(math.log(1.0 - confidence) / denom))) if denom < 0 else it


def estimate_affine_partial_2d(src_pts: np.ndarray, dst_pts: np.ndarray, confidence: float = 0.99,
                               reproj_threshold: float = 3.0, max_iters: int = 2000, seed: int = 0):
    """Counterpart of cv.estimateAffinePartial2D(src, dst, method=RANSAC, confidence=0.99): similarity transform
    (rotation, uniform scale, translation) mapping src to dst, robust to outliers.  Returns (2x3 matrix or None,
    inlier mask).

    This function is the DEFINITION the device path (Context.match_similarity -> ma_match_similarity, csrc/ransac.hip)
    reproduces bit for bit: samples are numpy's Generator(PCG64(seed)).choice(n, 2, replace=False) in sequence; a sample's
    model is the similarity through its two point pairs in closed form; inliers by _similarity_inliers; the iteration count
    adapts by ransac_iterations; the best sample's inliers are refitted by _fit_similarity and re-selected until the set
    is stable (at most 10 times).  OpenCV's own estimator has the same structure (two-point closed-form kernel, adaptive
    iteration count, refinement on the inliers) with its own random sequence and a Levenberg-Marquardt refinement."""
    src = np.asarray(src_pts, np.float64).reshape(-1, 2)
    dst = np.asarray(dst_pts, np.float64).reshape(-1, 2)
    n = len(src)
    if n < 2:
        return None, np.zeros(n, bool)
    rng = np.random.default_rng(seed)
    best_mask, best_count, iters, it = None, 0, max_iters, 0
    thr2 = reproj_threshold * reproj_threshold
    sx, sy, dx, dy = (np.ascontiguousarray(v) for v in (src[:, 0], src[:, 1], dst[:, 0], dst[:, 1]))
    while it < iters:
        it += 1
        i, j = rng.choice(n, 2, replace=False)
        # np.allclose(src[i], src[j]) spelled out (its call overhead was a third of a RANSAC round)
        if (abs(sx[i] - sx[j]) <= 1e-8 + 1e-5 * abs(sx[j])) and (abs(sy[i] - sy[j]) <= 1e-8 + 1e-5 * abs(sy[j])):
            continue
        # the similarity through two point pairs: z = (q1 - q0) / (p1 - p0) as complex numbers, t = q0 - z p0
        ux, uy, vx, vy = sx[j] - sx[i], sy[j] - sy[i], dx[j] - dx[i], dy[j] - dy[i]
        den = ux * ux + uy * uy
        if den == 0.0:
            continue
        a, b = (vx * ux + vy * uy) / den, (vy * ux - vx * uy) / den
        tx, ty = dx[i] - (a * sx[i] - b * sy[i]), dy[i] - (b * sx[i] + a * sy[i])
        ex, ey = a * sx - b * sy + tx - dx, b * sx + a * sy + ty - dy
        mask = ex * ex + ey * ey < thr2
        count = int(np.count_nonzero(mask))
        if count > best_count:
            best_count, best_mask = count, mask
            iters = ransac_iterations(count, n, confidence, max_iters, it)
    if best_mask is None or best_count < 2:
        return None, np.zeros(n, bool)
    M = _fit_similarity(src[best_mask], dst[best_mask])
    if M is None:
        return None, best_mask
    for _ in range(10):     # re-select the inliers of the refined model (OpenCV refines on the inlier set)
        mask = _similarity_inliers(M, sx, sy, dx, dy, thr2)
        if np.count_nonzero(mask) < 2 or np.array_equal(mask, best_mask):
            break
        best_mask = mask
        M2 = _fit_similarity(src[mask], dst[mask])
        if M2 is None:
            break
        M = M2
    return M, best_mask
