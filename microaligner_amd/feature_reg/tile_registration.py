"""Per-tile feature extraction and the image-pair fit (counterpart of
microaligner/feature_reg/tile_registration.py:27-97)."""
from typing import List

import numpy as np

from ..shared_modules.tiling import TileGrid
from .feature_detection import (TILE_OVERLAP, Features, find_features_device, find_features_parallelized,
                                match_features)


def split_image_into_tiles(img: np.ndarray, tile_size: int):
    """Zero-padded (tile_size + 2*51)^2 windows in row-major order + the slicer's info dict
    (tile_registration.py:27-34, slicer.py:69-118)."""
    grid = TileGrid(img.shape[0], img.shape[1], tile_size, TILE_OVERLAP)
    P = grid.window
    tiles = []
    for y0, x0 in grid.origins():
        win = np.zeros((P, P), img.dtype)
        ys, xs = max(y0, 0), max(x0, 0)
        ye, xe = min(y0 + P, img.shape[0]), min(x0 + P, img.shape[1])
        win[ys - y0:ye - y0, xs - x0:xe - x0] = img[ys:ye, xs:xe]
        tiles.append(win)
    return tiles, grid.slicer_info()


def combine_features(feature_list: List[Features], x_ntiles: int, y_ntiles: int, tile_size_x: int,
                     tile_size_y: int) -> Features:
    """Keypoints of every tile moved to image coordinates (tile origin + interior coordinate), descriptors
    concatenated in the same order (tile_registration.py:37-74)."""
    pts, responses, descriptors = [], [], []
    for tile_id, f in enumerate(feature_list):
        if not f.is_valid():
            continue
        origin = np.array([tile_id % x_ntiles * tile_size_x, tile_id // x_ntiles * tile_size_y], np.float64)
        pts.append(f.pts + origin)
        responses.append(f.responses)
        descriptors.append(f.descriptors)
    combined = Features()
    if pts:
        combined.pts = np.concatenate(pts, axis=0)
        combined.responses = np.concatenate(responses)
        combined.descriptors = np.concatenate(descriptors, axis=0)
    return combined


def find_features(img: np.ndarray, tile_size: int, ctx=None) -> Features:
    """Features of all tiles of `img` in image coordinates.  With a device context the dense work (FAST score map,
    DAISY layers / smoothing / sampling) runs there for all tiles at once, otherwise on a host thread per tile."""
    tiles, info = split_image_into_tiles(img, tile_size)
    tile_h, tile_w = info["tile_shape"]
    per_tile = find_features_device(tiles, ctx) if ctx is not None else find_features_parallelized(tiles)
    return combine_features(per_tile, info["ntiles"]["x"], info["ntiles"]["y"], tile_w, tile_h)


def register_img_pair(ref_combined_features: Features, mov_combined_features: Features, verbose: bool = True, knn=None, log=print,
                      ctx=None):
    return match_features(ref_combined_features, mov_combined_features, verbose, knn, log, ctx)
