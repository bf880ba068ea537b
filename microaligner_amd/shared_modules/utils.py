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
