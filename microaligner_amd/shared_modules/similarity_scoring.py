"""Mutual-information accept/reject gate (counterpart of
microaligner/shared_modules/similarity_scoring.py:27-68) on the HIP path.

The joint histograms and the per-chunk NMI run on the device (ma_nmi_u8); the host only takes
the mean of the chunk scores and compares, as the reference does.
"""
from typing import List, Tuple

import numpy as np

from ..device import DeviceArray, get_context
from .tiling import is_tiled


def _is_u8_or_all_zero(ctx, arr):
    """The gate's inputs are dog() outputs: uint8, or -- when dog() hit its `img.max() == 0` shortcut
    (optflow_registrator.py:256-257) -- the untouched image.  An all-zero image is a single-label image whatever its
    dtype and maps to the all-zero uint8 image; anything else that is not uint8 returns None (host path below)."""
    arr = ctx.asdevice(arr)
    if arr.dtype == np.uint8:
        return arr
    mn, mx = ctx.minmax(arr)
    if mn == 0 and mx == 0:
        return ctx.zeros(arr.shape, np.uint8)
    return None


def _nmi_of_labels(a: np.ndarray, b: np.ndarray) -> float:
    """sklearn.metrics.normalized_mutual_info_score(a, b) for arbitrary label values (metrics/cluster/_supervised.py:
    contingency table of the distinct values, mutual information with natural logarithms, arithmetic mean of the two
    entropies).  Host arithmetic for the one input class that has no device form: a float image whose maximum is 0 but
    which is not all zero (only non-positive values) comes out of the reference's dog() unchanged
    (optflow_registrator.py:256-257) and scikit-learn labels every distinct float value."""
    a, b = np.ravel(a), np.ravel(b)
    ua, ia = np.unique(a, return_inverse=True)
    ub, ib = np.unique(b, return_inverse=True)
    if len(ua) == 1 and len(ub) == 1:
        return 1.0
    n = float(a.size)
    pair, nij = np.unique(ia.astype(np.int64) * len(ub) + ib, return_counts=True)
    ai, bj = np.bincount(ia).astype(np.float64), np.bincount(ib).astype(np.float64)
    i, j = pair // len(ub), pair % len(ub)
    nij = nij.astype(np.float64)
    log_n = np.log(n)
    mi = nij / n * (np.log(nij) - log_n) + nij / n * (-np.log(ai[i] * bj[j]) + 2 * log_n)
    mi = np.where(np.abs(mi) < np.finfo(np.float64).eps, 0.0, mi).sum()
    mi = max(float(mi), 0.0)
    if abs(mi) < np.finfo(np.float64).eps:
        return 0.0

    def entropy(counts):
        if len(counts) == 1:
            return 0.0
        return float(-np.sum(counts / n * (np.log(counts) - log_n)))

    norm = max(0.5 * (entropy(ai) + entropy(bj)), np.finfo(np.float64).eps)
    return mi / norm


def mi_tiled(arr1, arr2, tile_size: int) -> float:
    """similarity_scoring.py:27-50: one score if max(shape)/tile_size < 2, else the mean over
    consecutive runs of tile_size**2 elements of the flattened arrays."""
    ctx = get_context()
    a, b = _is_u8_or_all_zero(ctx, arr1), _is_u8_or_all_zero(ctx, arr2)
    shape = np.shape(arr1) if not isinstance(arr1, DeviceArray) else arr1.shape
    chunk = tile_size * tile_size if is_tiled(shape, tile_size) else 0
    if a is None or b is None:
        # raw (non-uint8, not all-zero) labels: scikit-learn's definition on the host, same chunking as the reference
        ha = np.ravel(arr1.numpy() if isinstance(arr1, DeviceArray) else np.asarray(arr1))
        hb = np.ravel(arr2.numpy() if isinstance(arr2, DeviceArray) else np.asarray(arr2))
        if chunk == 0:
            return _nmi_of_labels(ha, hb)
        return np.mean([_nmi_of_labels(ha[s:s + chunk], hb[s:s + chunk]) for s in range(0, ha.size, chunk)])
    scores = ctx.nmi_scores(a, b, chunk)
    if chunk == 0:
        return float(scores[0])
    return np.mean(scores)


def mutual_information_test(ref_arr, test_arr, init_arr, tile_size: int) -> Tuple[float, float]:
    if all(isinstance(v, DeviceArray) and v.dtype == np.uint8 for v in (ref_arr, test_arr, init_arr)) \
            and ref_arr.shape == test_arr.shape == init_arr.shape:
        # the usual case (three dog() outputs on the device): the two scores share the reference labels, one pair of launches
        # and one synchronisation (ma_nmi_u8_pair); the same kernels on the same inputs as two mi_tiled calls
        ctx = get_context()
        chunk = tile_size * tile_size if is_tiled(ref_arr.shape, tile_size) else 0
        s_after, s_before = ctx.nmi_scores_pair(ref_arr, test_arr, init_arr, chunk)
        if chunk == 0:
            return float(s_after[0]), float(s_before[0])
        return np.mean(s_after), np.mean(s_before)
    after_mi_score = mi_tiled(ref_arr, test_arr, tile_size)
    before_mi_score = mi_tiled(ref_arr, init_arr, tile_size)
    return after_mi_score, before_mi_score


def check_if_higher_similarity(ref_arr, test_arr, init_arr, tile_size: int, verbose: bool = True, log=print) -> List[bool]:
    after, before = mutual_information_test(ref_arr, test_arr, init_arr, tile_size)
    if verbose:
        log("    MI score after:", after, "| MI score before:", before)
    return [after > before]
