"""Fixtures for transform_img_with_tmat from the REAL scikit-image (0.18.3) of the build container:

    /opt/conda/bin/python3.9 tests/golden/make_affine_golden.py

Re-implements nothing: it calls skimage.transform.warp exactly as microaligner/shared_modules/utils.py:98-114
does (pad -> pinv -> AffineTransform -> warp(preserve_range=True) -> astype) and stores inputs, matrices and
outputs in tests/golden/affine_cases.npz (inputs are stored too: numpy/scipy versions differ between interpreters).
"""
import os
import warnings

import numpy as np

warnings.filterwarnings("ignore")
from skimage.transform import AffineTransform, warp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def pad_to_shape(img, target_shape):
    if tuple(img.shape) == tuple(target_shape):
        return img
    def split(t, a):
        d = t - a
        return (0, 0) if d <= 0 else (d // 2, d - d // 2)
    (l, r), (t, b) = split(target_shape[1], img.shape[1]), split(target_shape[0], img.shape[0])
    return np.pad(img, ((t, b), (l, r)), mode="constant")


def reference(img, target_shape, tmat):
    """Returns (output, inverse matrix handed to skimage) -- pinv depends on the LAPACK build in its last bits and
    near-integer coordinates amplify that into +-1 differences, so the fixture records the matrix actually used."""
    dtype = img.dtype
    img = pad_to_shape(img, target_shape)
    inv = np.linalg.pinv(np.append(tmat, [[0, 0, 1]], axis=0))
    if np.array_equal(tmat, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])):
        return img, inv
    return warp(img, AffineTransform(inv), output_shape=img.shape, preserve_range=True).astype(dtype), inv


def rot(deg, tx, ty, s=1.0):
    t = np.deg2rad(deg)
    return np.array([[s * np.cos(t), -s * np.sin(t), tx], [s * np.sin(t), s * np.cos(t), ty]])


rng = np.random.default_rng(7)
cases = {}
specs = [("u16_rot", np.uint16, (61, 83), (61, 83), rot(0.4, 3.3, -2.1)),
         ("u8_rot_pad", np.uint8, (50, 70), (64, 90), rot(-1.5, -4.25, 6.5)),
         ("f32_rot", np.float32, (72, 64), (72, 64), rot(0.7, 1.25, 2.5, 1.01)),
         ("u16_shift_int", np.uint16, (40, 48), (40, 48), np.array([[1.0, 0.0, 5.0], [0.0, 1.0, -3.0]])),
         ("u16_shift_frac", np.uint16, (40, 48), (40, 48), np.array([[1.0, 0.0, 0.5], [0.0, 1.0, 0.25]])),
         ("u8_identity_pad", np.uint8, (30, 31), (33, 36), np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])),
         ("u16_lowrange", np.uint16, (45, 45), (45, 45), rot(2.0, 0.0, 0.0)),
         ("f32_negative", np.float32, (33, 57), (33, 57), rot(-0.9, 2.0, 1.0))]
for name, dt, shape, target, tmat in specs:
    if dt == np.float32:
        img = (rng.standard_normal(shape) * 100).astype(np.float32) if "negative" in name else \
            (rng.random(shape) * 255).astype(np.float32)
    elif "lowrange" in name:
        img = (rng.integers(1000, 1100, shape)).astype(dt)   # min > 0: the cval-preserving clip branch
    else:
        img = (rng.random(shape) * np.iinfo(dt).max).astype(dt)
    out, inv = reference(img, target, tmat)
    cases[name + "__inv"] = inv
    cases[name + "__img"] = img
    cases[name + "__tmat"] = tmat
    cases[name + "__target"] = np.array(target)
    cases[name + "__out"] = out
    print(name, out.dtype, out.shape)
np.savez_compressed(os.path.join(HERE, "affine_cases.npz"), **cases)
