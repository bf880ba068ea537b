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


def max_project_and_normalize(pages, on_device: bool = False):
    """In-memory counterpart of read_and_max_project_pages (utils.py:75-95): element-wise maximum over the z pages
    (np.maximum fold, :92) followed by cv2.normalize(..., 0, 255, NORM_MINMAX, CV_8U) (:94), both on the device.
    `pages`: a (Z, H, W) array or a sequence of (H, W) pages (uint8 / uint16 / float32).  Returns uint8."""
    from ..device import DeviceArray, get_context
    ctx = get_context()
    if isinstance(pages, DeviceArray):
        stack = pages
    else:
        stack = ctx.asdevice(np.ascontiguousarray(np.stack([np.asarray(p) for p in pages])))
    if stack.ndim != 3:
        raise ValueError(f"expected (Z, H, W) pages, got shape {stack.shape}")
    proj = ctx.max_project(stack) if stack.shape[0] > 1 else DeviceArray(ctx, stack.shape[1:], stack.dtype, stack.ptr, 0, owner=False)
    out = ctx.normalize_minmax_u8(proj)
    del proj
    return out if on_device else out.numpy()
