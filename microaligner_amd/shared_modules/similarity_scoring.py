"""Mutual-information accept/reject gate (counterpart of
microaligner/shared_modules/similarity_scoring.py:27-68) on the HIP path.

The joint histograms and the per-chunk NMI run on the device (ma_nmi_u8); the host only takes
the mean of the chunk scores and compares, as the reference does.
"""
from typing import List, Tuple

import numpy as np

from ..device import DeviceArray, get_context
from .tiling import is_tiled


def _as_u8_labels(ctx, arr):
    """The gate's inputs are dog() outputs: uint8, or -- when dog() hit its `img.max() == 0`
    shortcut (optflow_registrator.py:256-257) -- the untouched image.  An all-zero image is a
    single-label image whatever its dtype; anything else that is not uint8 has no counterpart
    on the device (sklearn would label every distinct float value)."""
    arr = ctx.asdevice(arr)
    if arr.dtype == np.uint8:
        return arr
    mn, mx = ctx.minmax(arr)
    if mn == 0 and mx == 0:
        return ctx.zeros(arr.shape, np.uint8)
    raise NotImplementedError(
        "NMI gate received a non-uint8 image that is not all zero (dog() returned its input "
        "unchanged because img.max() == 0 on an image with negative values); not supported")


def mi_tiled(arr1, arr2, tile_size: int) -> float:
    """similarity_scoring.py:27-50: one score if max(shape)/tile_size < 2, else the mean over
    consecutive runs of tile_size**2 elements of the flattened arrays."""
    ctx = get_context()
    a, b = _as_u8_labels(ctx, arr1), _as_u8_labels(ctx, arr2)
    chunk = tile_size * tile_size if is_tiled(a.shape, tile_size) else 0
    scores = ctx.nmi_scores(a, b, chunk)
    if chunk == 0:
        return float(scores[0])
    return np.mean(scores)


def mutual_information_test(ref_arr, test_arr, init_arr, tile_size: int) -> Tuple[float, float]:
    after_mi_score = mi_tiled(ref_arr, test_arr, tile_size)
    before_mi_score = mi_tiled(ref_arr, init_arr, tile_size)
    return after_mi_score, before_mi_score


def check_if_higher_similarity(ref_arr, test_arr, init_arr, tile_size: int, verbose: bool = True) -> List[bool]:
    after, before = mutual_information_test(ref_arr, test_arr, init_arr, tile_size)
    if verbose:
        print("    MI score after:", after, "| MI score before:", before)
    return [after > before]
