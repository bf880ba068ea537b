"""2x3 affine bookkeeping of the feature-based registration (homogeneous products, translation rescaling between
pyramid levels, plausibility checks).  Behavioural counterpart of the private helpers of
microaligner/feature_reg/feature_registrator.py:214-279, derived from the geometry rather than from that code."""
from math import hypot

import numpy as np

IDENTITY = np.eye(2, 3)
SCALE_RANGE = (0.3, 3.0)   # a level's estimate may shrink / stretch an axis by at most this much


def homogeneous(t_mat):
    return np.vstack([np.asarray(t_mat, dtype=np.float64), [0.0, 0.0, 1.0]])


def compose(mats):
    """Product of the transforms as homogeneous matrices, first matrix leftmost; a single matrix is returned as is."""
    mats = list(mats)
    if len(mats) == 1:
        return mats[0]
    acc = homogeneous(mats[0])
    for m in mats[1:]:
        acc = acc @ homogeneous(m)
    return acc[:2]


def with_translation_scaled(t_mat, factor):
    """The same linear part with the translation expressed in pixels of a level `factor` times finer."""
    out = np.array(t_mat, dtype=np.float64, copy=True)
    out[:, 2] *= factor
    return out


def axis_scales(t_mat):
    """Stretch factors of the linear part from its Gram-Schmidt (QR) factorisation: the length of the leading
    non-zero column and the signed area of the unit square's image divided by that length.  None for the zero map."""
    lin = np.asarray(t_mat, dtype=np.float64)[:, :2]
    area = lin[0, 0] * lin[1, 1] - lin[1, 0] * lin[0, 1]
    first, second = hypot(lin[0, 0], lin[1, 0]), hypot(lin[0, 1], lin[1, 1])
    if first > 0:
        return first, area / first
    if second > 0:
        return area / second, second
    return None


def scales_plausible(t_mat):
    scales = axis_scales(t_mat)
    return scales is not None and all(SCALE_RANGE[0] <= abs(s) <= SCALE_RANGE[1] for s in scales)


def centre_stays_inside(t_mat, img_shape):
    """The image centre, transformed, must not leave the image extent (|x'| <= width, |y'| <= height)."""
    h, w = img_shape[:2]
    moved = homogeneous(t_mat) @ np.array([w // 2, h // 2, 1.0])
    return bool(abs(moved[0]) <= w and abs(moved[1]) <= h)
