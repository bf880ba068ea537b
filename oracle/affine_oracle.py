"""CPU oracle of transform_img_with_tmat (microaligner/shared_modules/utils.py:98-114), i.e. of
skimage.transform.warp(img, AffineTransform(pinv(M3x3)), output_shape=img.shape, preserve_range=True)
.astype(original dtype) with bilinear interpolation, constant border 0 and clipping to the input range.

TEST INFRASTRUCTURE.  Unlike the OpenCV primitives this one IS pinned: scikit-image 0.18.3 exists in the build
container (/opt/conda/bin/python3.9) and tests/golden/make_affine_golden.py stores its outputs as fixtures;
tests/test_oracle_affine.py requires this restatement to reproduce them bit for bit.

Restated from skimage/transform/_warps.py:684-250 (warp), _warps_cy.pyx (_warp_fast: per output pixel
transform (col, row) with the 3x3 matrix -- metric / affine / projective branch chosen by exact comparisons on
the matrix entries -- then bilinear_interpolation with floor/ceil corners) and _clip_warp_output.
float32 images are processed in float32, every other dtype in float64 (convert_to_float).
"""
import numpy as np


def pad_to_shape(img, target_shape):
    if tuple(img.shape) == tuple(target_shape):
        return img
    def split(t, a):
        d = t - a
        return (0, 0) if d <= 0 else (d // 2, d - d // 2)
    (l, r), (t, b) = split(target_shape[1], img.shape[1]), split(target_shape[0], img.shape[0])
    return np.pad(img, ((t, b), (l, r)), mode="constant")


def inverse_matrix(tmat):
    """np.linalg.pinv of the homogeneous 3x3 matrix, as the reference computes it (utils.py:107-109)."""
    return np.linalg.pinv(np.append(np.asarray(tmat, dtype=np.float64), [[0, 0, 1]], axis=0))


def warp_with_inverse(img, inv):
    ft = np.float32 if img.dtype == np.float32 else np.float64
    H, W = img.shape
    im = img.astype(ft)
    M = inv.astype(ft).ravel()
    r, c = np.mgrid[0:H, 0:W]
    r, c = r.astype(ft), c.astype(ft)
    if M[6] == 0 and M[7] == 0 and M[8] == 1:
        if M[1] == 0 and M[3] == 0:
            x = M[0] * c + M[2]
            y = M[4] * r + M[5]
        else:
            x = M[0] * c + M[1] * r + M[2]
            y = M[3] * c + M[4] * r + M[5]
    else:
        z = M[6] * c + M[7] * r + M[8]
        x = (M[0] * c + M[1] * r + M[2]) / z
        y = (M[3] * c + M[4] * r + M[5]) / z
    minr, minc = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
    maxr, maxc = np.ceil(y).astype(np.int64), np.ceil(x).astype(np.int64)
    dr, dc = (y - minr.astype(ft)).astype(ft), (x - minc.astype(ft)).astype(ft)

    def px(rr, cc):
        ok = (rr >= 0) & (rr < H) & (cc >= 0) & (cc < W)
        v = np.zeros((H, W), ft)
        v[ok] = im[rr[ok], cc[ok]]
        return v

    one = ft(1)
    top = (one - dc) * px(minr, minc) + dc * px(minr, maxc)
    bot = (one - dc) * px(maxr, minc) + dc * px(maxr, maxc)
    out = ((one - dr) * top + dr * bot).astype(ft)
    lo, hi = im.min(), im.max()
    keep = (out == 0) if not (lo <= 0 <= hi) else None
    out = np.clip(out, lo, hi)
    if keep is not None:
        out[keep] = 0
    return out


def transform_img_with_tmat(img, target_shape, tmat):
    dtype = img.dtype
    img = pad_to_shape(img, target_shape)
    if np.array_equal(tmat, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])):
        return img
    with np.errstate(invalid="ignore"):
        return warp_with_inverse(img, inverse_matrix(tmat)).astype(dtype)
