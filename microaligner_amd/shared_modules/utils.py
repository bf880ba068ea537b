"""Helpers exported next to the registrators (counterpart of microaligner/shared_modules/utils.py)."""
from typing import Tuple

import numpy as np


def _split_padding(target: int, actual: int) -> Tuple[int, int]:
    extra = target - actual
    if extra <= 0:
        return 0, 0
    before = extra // 2
    return before, extra - before


def pad_to_shape(img: np.ndarray, target_shape: Tuple[int, int]):
    """Centre `img` in a zero canvas of `target_shape`; returns (padded, (left, right, top, bottom))
    like utils.py:53-66 (cv2.copyMakeBorder with BORDER_CONSTANT 0)."""
    if tuple(img.shape) == tuple(target_shape):
        return img, (0, 0, 0, 0)
    left, right = _split_padding(target_shape[1], img.shape[1])
    top, bottom = _split_padding(target_shape[0], img.shape[0])
    padded = np.pad(img, ((top, bottom), (left, right)), mode="constant")
    return padded, (left, right, top, bottom)


def transform_img_with_tmat(img, target_shape: Tuple[int, int], transform_matrix: np.ndarray):
    """Pad `img` to `target_shape` and apply the 2x3 affine `transform_matrix` (utils.py:98-114): the identity
    matrix is a no-op, anything else is skimage.transform.warp with the pseudo-inverse of the homogeneous matrix
    (bilinear, constant border 0, preserve_range, clipped to the input range) cast back to the input dtype -- here
    one HIP kernel (ma_warp_affine).  Integer dtypes reproduce scikit-image bit for bit given the same inverse
    matrix; float32 images agree to 2 ulp (scikit-image's float32 code path is not pinned)."""
    from ..device import DeviceArray, get_context
    original_dtype = img.dtype
    img, _ = pad_to_shape(img, target_shape)
    identity = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    if np.array_equal(transform_matrix, identity):
        return img
    inv = np.linalg.pinv(np.append(np.asarray(transform_matrix, dtype=np.float64), [[0, 0, 1]], axis=0))
    ctx = get_context()
    out = ctx.warp_affine(ctx.asdevice(np.ascontiguousarray(img)), inv)
    return out.numpy().astype(original_dtype, copy=False)
