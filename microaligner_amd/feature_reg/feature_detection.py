"""Keypoints, descriptors and their matching for FeatureRegistrator.

Counterpart of microaligner/feature_reg/feature_detection.py (Features :27-85, find_features :88-120,
match_features :123-158, find_features_parallelized :161-168).  The reference gets FAST, DAISY, FLANN and the
RANSAC similarity fit from opencv-contrib; here they come from sparse_cpu.py (see its header for what is and is not
the same).  Tiles are processed by a thread pool instead of dask.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence

import numpy as np

from .sparse_cpu import Daisy, KeyPoint, estimate_affine_partial_2d, fast_detect, knn2

TILE_OVERLAP = 51            # overlap of the feature tiles (tile_registration.py:31) and the margin FAST skips
RATIO = 0.5                  # Lowe's ratio (feature_detection.py:143)


class Features:
    """Keypoints + descriptors of one image (or tile).  `keypoints` is a list of KeyPoint (pt, size, angle,
    response, octave, class_id -- the cv2.KeyPoint fields the reference serialises) or None.  The device path keeps
    the points as arrays (`pts` (n, 2) float64 (x, y), `responses` (n,)) and builds the KeyPoint list only when somebody
    asks for it: a level of a 4096^2 image has ~45 000 of them per image and iteration.  Features that
    ma_feature_extract made live on the device altogether (descriptors, points, responses): `pts`, `responses` and
    `descriptors` download them when asked for, the matching step (ma_knn2_l2, ma_match_similarity) never does."""

    def __init__(self):
        self._keypoints: Optional[List[KeyPoint]] = None
        self._pts: Optional[np.ndarray] = None
        self._responses: Optional[np.ndarray] = None
        self._descriptors = None           # ndarray (n, 200) float32, or a DeviceArray that is downloaded when asked for
        self._dev = None                   # (ctx, points buffer, responses buffer, n) of device-made features
        self._pts_dev = None

    @classmethod
    def on_device(cls, ctx, desc, pts, resp, n):
        f = cls()
        f._descriptors, f._dev, f._pts_dev = desc, (ctx, pts, resp, n), pts
        return f

    @classmethod
    def pending_on_device(cls, ctx, desc, pts, resp, word):
        """Features whose extraction has been ENQUEUED (Context.feature_extract(wait=False)): `word` receives the keypoint count
        in stream order.  settle() -- after the caller has waited for an event recorded behind the extraction -- turns the
        object into what on_device() (or, without keypoints, Features()) would have made."""
        f = cls.on_device(ctx, desc, pts, resp, None)
        f._pending = word
        return f

    def settle(self):
        word = getattr(self, "_pending", None)
        if word is None:
            return self
        self._pending = None
        n = int(word[0])
        ctx, pts, resp, _ = self._dev
        if n == 0:
            self._descriptors, self._dev, self._pts_dev = None, None, None
        else:
            self._descriptors.shape = (n, 200)
            self._dev = (ctx, pts, resp, n)
        return self

    @property
    def pts(self) -> Optional[np.ndarray]:
        if self._pts is None and self._dev is not None:
            ctx, pts, _, n = self._dev
            self._pts = ctx.download_raw(pts, (n, 2), np.float64)
        return self._pts

    @pts.setter
    def pts(self, v):
        self._pts, self._pts_dev = v, None

    @property
    def responses(self) -> Optional[np.ndarray]:
        if self._responses is None and self._dev is not None:
            ctx, _, resp, n = self._dev
            self._responses = ctx.download_raw(resp, (n,), np.int32).astype(np.float64)
        return self._responses

    @responses.setter
    def responses(self, v):
        self._responses = v

    @property
    def descriptors(self) -> Optional[np.ndarray]:
        if self._descriptors is not None and not isinstance(self._descriptors, np.ndarray):
            self._descriptors = self._descriptors.numpy()
        return self._descriptors

    @descriptors.setter
    def descriptors(self, des):
        self._descriptors = des

    @property
    def descriptors_for_search(self):
        """The descriptors where they are (host array or DeviceArray): what the 2-NN search takes."""
        return self._descriptors

    @property
    def keypoints(self) -> Optional[List[KeyPoint]]:
        if self._keypoints is None and self.pts is not None:
            self._keypoints = [KeyPoint((float(x), float(y)), 7.0, -1.0, float(r), 0, -1)
                               for (x, y), r in zip(self.pts, self.responses)]
        return self._keypoints

    @keypoints.setter
    def keypoints(self, kps):
        self._keypoints = kps
        if kps is not None:
            self.pts = np.array([k.pt for k in kps], np.float64).reshape(-1, 2)
            self.responses = np.array([k.response for k in kps], np.float64)
        else:
            self.pts = self.responses = None

    def is_valid(self) -> bool:
        return (self._pts is not None or self._dev is not None) and self._descriptors is not None

    def device_pts(self, ctx):
        """The keypoint positions as an (n, 2) float64 device buffer (uploaded once per Features object)."""
        if self._pts_dev is None:
            self._pts_dev = ctx._upload_raw(np.ascontiguousarray(self.pts, np.float64))
        return self._pts_dev


def view_tile_without_overlap(img, overlap):
    return img[overlap:-overlap, overlap:-overlap]


def find_features(img: np.ndarray, nfeatures_limit: int = 5000) -> Features:
    """feature_detection.py:88-120.  FAST corners of the tile's interior (the overlap margin is cut off first, so
    keypoint coordinates are relative to the interior), strongest `nfeatures_limit` first, DAISY descriptors
    computed on the FULL tile at those coordinates -- i.e. 51 px up-left of the corner, the reference's own
    convention (:105 vs :107), kept because reference and moving image are treated alike."""
    img = np.asarray(img)
    features = Features()
    if img.size == 0 or img.max() == 0:
        return features
    kp = fast_detect(view_tile_without_overlap(img, TILE_OVERLAP), threshold=1, nonmax=True)
    kp = sorted(kp, key=lambda k: k.response, reverse=True)[:nfeatures_limit]
    des = Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8).compute(img, kp)
    if len(kp) < 3 or des is None or len(des) < 3:
        return features
    features.keypoints, features.descriptors = kp, des
    return features


def match_features(img1_features: Features, img2_features: Features, verbose: bool = True, knn=None, log=print, ctx=None) -> np.ndarray:
    """feature_detection.py:123-158: 2-NN of every descriptor of image 2 among those of image 1, ratio test, then
    the similarity transform that maps image-2 points onto image-1 points.  Identity when there is too little to
    go on; None (as cv2 does) when the fit itself fails is mapped to identity as well.
    knn: the 2-NN search, (query, train) -> (idx, dist); the default is the host one (sparse_cpu.knn2).
    ctx: a device context -- the whole step runs there (Context.knn2 -> ma_knn2_l2 leaves its pairs on the device,
    Context.match_similarity -> ma_match_similarity does the ratio test and the RANSAC fit on them, bit for bit the host
    statement below); only the matrix and the number of good matches come back."""
    identity = np.eye(2, 3)
    if not img1_features.is_valid() or not img2_features.is_valid():
        return identity
    if ctx is not None:
        des1, des2 = img1_features.descriptors_for_search, img2_features.descriptors_for_search
        if len(des1) < 2:
            return identity
        idx_d, dist_d = ctx.knn2(des2, des1, on_device=True)
        mat, n_good, status = ctx.match_similarity(idx_d, dist_d, img2_features.device_pts(ctx), img1_features.device_pts(ctx),
                                                   ratio=RATIO, confidence=0.99)
        if status != 3:
            if verbose:
                log("    Good matches", n_good, "/", len(des2))
            return identity if mat is None else mat
        knn = ctx.knn2            # coordinates the device path does not take (not integer-valued): the host statement decides
    search = knn or knn2
    pts1, pts2 = img1_features.pts, img2_features.pts
    if knn is not None:     # the device search takes the descriptors where they are
        des1, des2 = img1_features.descriptors_for_search, img2_features.descriptors_for_search
    else:
        des1, des2 = img1_features.descriptors, img2_features.descriptors
    if len(des1) < 2:
        return identity
    idx, dist = search(des2, des1)
    good = np.nonzero(dist[:, 0] < RATIO * dist[:, 1])[0]
    if verbose:
        log("    Good matches", len(good), "/", len(des2))   # log: where the line goes (print, or a registrator's buffer)
    if len(good) < 3:
        return identity
    src_pts = pts1[idx[good, 0]].astype(np.float32)
    dst_pts = pts2[good].astype(np.float32)
    mat, _ = estimate_affine_partial_2d(dst_pts, src_pts, confidence=0.99)
    return identity if mat is None else mat


_DAISY_TABLES = {}


def _daisy_tables(daisy: Daisy):
    """Host-side constants of the device DAISY (ma_daisy_describe): the centre-first halves of scipy's Gaussian
    kernels for the three smoothing increments (truncate 3.0), the (cos, sin) of the orientation bins and the
    sampling offsets -- the very doubles sparse_cpu.Daisy uses.  Computed once per parameter set (a register() asks
    for them twelve times)."""
    key = (daisy.radius, daisy.q_radius, daisy.q_theta, daisy.q_hist)
    if key not in _DAISY_TABLES:
        _DAISY_TABLES[key] = _daisy_tables_uncached(daisy)
    return _DAISY_TABLES[key]


def _daisy_tables_uncached(daisy: Daisy):
    from scipy.ndimage._filters import _gaussian_kernel1d
    halves = []
    for inc in daisy.smoothing_increments():
        lw = int(3.0 * inc + 0.5)
        halves.append(_gaussian_kernel1d(inc, 0, lw)[lw:])
    th = [2.0 * np.pi * o / daisy.q_hist for o in range(daisy.q_hist)]
    cos_sin = np.array([(np.cos(t), np.sin(t)) for t in th], np.float64)
    return halves, cos_sin, daisy.sample_offsets()


DEVICE_WORKSPACE_BYTES = 8 << 30   # budget for the DAISY cubes of one batch of tiles (4 cubes x 8 planes x P^2 float32 each)


def _device_batches(n_tiles: int, P: int, budget: int) -> List[range]:
    """Tile index ranges whose DAISY workspace (128 P^2 bytes per tile) fits `budget` and whose plane count fits a grid
    dimension (8 planes per tile, 65535 blocks along z): a 25 000^2 level at tile_size 200 is 15 876 tiles."""
    per_tile = 4 * 8 * P * P * 4
    step = max(1, min(budget // per_tile, 65535 // 8))
    return [range(s, min(s + step, n_tiles)) for s in range(0, n_tiles, step)]


def find_features_device(tile_list: Sequence[np.ndarray], ctx, workspace_bytes: Optional[int] = None) -> List[Features]:
    """find_features_parallelized with the dense work on the device, the tiles of the level in batches sized from a
    workspace budget: the FAST score map with non-maximum suppression (ma_fast_nms) and the DAISY orientation layers,
    their three Gaussian smoothings and the descriptor sampling (ma_daisy_describe).  The host keeps what is sparse:
    picking the strongest corners of each tile.  Same keypoints, same descriptors as find_features (tests compare them);
    like the reference, any image size works -- large levels just take more batches."""
    n_tiles = len(tile_list)
    if n_tiles == 0:
        return []
    limit = min(1000000 // n_tiles, 5000)
    feats = [Features() for _ in range(n_tiles)]
    P = int(np.shape(tile_list[0])[0])
    if P <= 2 * TILE_OVERLAP:
        return feats
    daisy = Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8)
    halves, cos_sin, offsets = _daisy_tables(daisy)
    budget = DEVICE_WORKSPACE_BYTES if workspace_bytes is None else int(workspace_bytes)
    for batch in _device_batches(n_tiles, P, budget):
        tiles = np.ascontiguousarray(np.stack([np.asarray(tile_list[t]) for t in batch]))
        if tiles.dtype != np.uint8:
            raise ValueError("FAST works on uint8 images (the DOG output)")
        d_tiles = ctx.asdevice(tiles)
        score = ctx.fast_nms(d_tiles, TILE_OVERLAP, threshold=1)
        picked, kp_tile, kp_xy = {}, [], []
        for k, t in enumerate(batch):
            if tiles[k].max() == 0:
                continue
            ys, xs = np.nonzero(score[k])
            if len(ys) == 0:
                continue
            resp = score[k][ys, xs]
            order = np.argsort(-resp, kind="stable")[:limit]      # strongest first, row-major order among equals
            picked[t] = (xs[order], ys[order], resp[order])
            kp_tile.append(np.full(len(order), k, np.int32))
            kp_xy.append(np.stack([xs[order], ys[order]], 1).astype(np.float64))
        if not picked:
            continue
        des = ctx.daisy_describe(d_tiles, np.concatenate(kp_tile), np.concatenate(kp_xy), halves, cos_sin, offsets)
        pos = 0
        for t, (xs, ys, resp) in picked.items():
            n = len(xs)
            if n >= 3:
                feats[t].pts = np.stack([xs, ys], 1).astype(np.float64)
                feats[t].responses = resp.astype(np.float64)
                feats[t].descriptors = des[pos:pos + n]
            pos += n
    return feats


def find_features_of_device_image(img, tile_size: int, ctx, workspace_bytes: Optional[int] = None, wait: bool = True):
    """tile_registration.find_features for a uint8 image that is already on the device (the DOG output), in one call
    (ma_feature_extract): the feature windows are cut there, the corners detected, ranked and cut to the per-tile limit
    there, the selection compacted in combine_features' layout (tile by tile, strongest first within a tile, tiles with
    fewer than three keypoints dropped, image coordinates) and the descriptors computed there -- and everything is LEFT
    there for the matching step; the number of keypoints is all that comes back.  Returns the combined Features; their
    `pts` / `responses` / `descriptors` download on demand."""
    from ..shared_modules.tiling import TileGrid
    H, W = img.shape
    grid = TileGrid(H, W, tile_size, TILE_OVERLAP)
    n_tiles, P = grid.ntiles, grid.window
    limit = min(1000000 // n_tiles, 5000)
    if P <= 2 * TILE_OVERLAP or limit < 1:
        return Features()
    daisy = Daisy(radius=21, q_radius=3, q_theta=8, q_hist=8)
    halves, cos_sin, offsets = _daisy_tables(daisy)
    budget = DEVICE_WORKSPACE_BYTES if workspace_bytes is None else int(workspace_bytes)
    desc, pts, resp, n = ctx.feature_extract(img, tile_size, TILE_OVERLAP, limit, halves, cos_sin, offsets, threshold=1,
                                             workspace_bytes=budget, wait=wait)
    if not wait:          # enqueued: the count arrives in stream order (Features.settle)
        return Features.pending_on_device(ctx, desc, pts, resp, n)
    if n == 0:
        return Features()
    return Features.on_device(ctx, desc, pts, resp, n)


def find_features_parallelized(tile_list: Sequence[np.ndarray], workers: Optional[int] = None) -> List[Features]:
    """feature_detection.py:161-168: at most 1 000 000 features over all tiles, at most 5000 per tile."""
    n_tiles = len(tile_list)
    if n_tiles == 0:
        return []
    limit = min(1000000 // n_tiles, 5000)
    if n_tiles == 1:
        return [find_features(tile_list[0], limit)]
    if workers is None:   # the numpy / scipy passes of a tile release the GIL: one host thread per tile, up to 32
        workers = min(32, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=min(workers, n_tiles)) as ex:
        return list(ex.map(lambda t: find_features(t, limit), tile_list))
