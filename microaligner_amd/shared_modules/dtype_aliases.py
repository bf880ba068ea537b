"""Type aliases used in signatures (counterpart of microaligner/shared_modules/dtype_aliases.py:24-42)."""
from typing import Tuple

import numpy as np

Image = np.ndarray          # 2-D grayscale image
Flow = np.ndarray           # (H, W, 2) float32, [..., 0] = dx, [..., 1] = dy
TMat = np.ndarray           # 2x3 affine matrix
Shape2D = Tuple[int, int]
Padding = Tuple[int, int, int, int]  # left, right, top, bottom
XML = str
