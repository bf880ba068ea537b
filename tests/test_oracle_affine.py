"""transform_img_with_tmat: the numpy oracle (oracle/affine_oracle.py) against fixtures produced by the REAL
scikit-image 0.18.3 (tests/golden/make_affine_golden.py), and the HIP kernel against both (-m gpu)."""
import os

import numpy as np
import pytest

from oracle import affine_oracle as A

CASES = np.load(os.path.join(os.path.dirname(__file__), "golden", "affine_cases.npz"))
NAMES = sorted({k.split("__")[0] for k in CASES.files})


def oracle_with_fixture_inverse(name):
    """The fixture's own inverse matrix (pinv differs between LAPACK builds in its last bits)."""
    img, tmat, target = CASES[name + "__img"], CASES[name + "__tmat"], tuple(CASES[name + "__target"])
    padded = A.pad_to_shape(img, target)
    if np.array_equal(tmat, np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])):
        return padded
    return A.warp_with_inverse(padded, CASES[name + "__inv"]).astype(img.dtype)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_real_skimage(name):
    out = oracle_with_fixture_inverse(name)
    ref = CASES[name + "__out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if ref.dtype == np.float32:
        np.testing.assert_allclose(out, ref, rtol=3e-7, atol=1e-4)    # skimage's f32 path is not bit-pinned
    else:
        assert np.array_equal(out, ref)                                # integer dtypes: bit for bit


def test_public_helper_semantics():
    img, tmat, target = CASES["u8_identity_pad__img"], CASES["u8_identity_pad__tmat"], (33, 36)
    out = A.transform_img_with_tmat(img, target, tmat)
    assert out.shape == target and out.sum() == img.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_warp_affine_matches_oracle_bit_for_bit(name):
    import microaligner_amd as ma
    img, tmat, target = CASES[name + "__img"], CASES[name + "__tmat"], tuple(CASES[name + "__target"])
    got = ma.transform_img_with_tmat(img, target, tmat)
    exp = A.transform_img_with_tmat(img, target, tmat)
    assert got.dtype == img.dtype and got.shape == exp.shape
    assert np.array_equal(got, exp)


@pytest.mark.gpu
def test_hip_warp_affine_large_u16():
    import microaligner_amd as ma
    rng = np.random.default_rng(1)
    img = rng.integers(0, 65535, (700, 900)).astype(np.uint16)
    t = np.deg2rad(0.3)
    tmat = np.array([[np.cos(t), -np.sin(t), 12.5], [np.sin(t), np.cos(t), -7.25]])
    assert np.array_equal(ma.transform_img_with_tmat(img, (720, 930), tmat), A.transform_img_with_tmat(img, (720, 930), tmat))


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in NAMES if "identity" not in n])
def test_hip_warp_affine_reproduces_real_skimage_fixtures(name, ctx):
    img, target = CASES[name + "__img"], tuple(CASES[name + "__target"])
    padded = np.ascontiguousarray(A.pad_to_shape(img, target))
    got = ctx.warp_affine(ctx.asdevice(padded), CASES[name + "__inv"]).numpy()
    ref = CASES[name + "__out"]
    if ref.dtype == np.float32:
        np.testing.assert_allclose(got, ref, rtol=3e-7, atol=1e-4)
    else:
        assert np.array_equal(got, ref)
