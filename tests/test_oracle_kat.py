"""Known-answer tests that pin the CPU oracle -- no OpenCV required.

The OpenCV primitives are 'parity unpinned' (no cv2 in this image, no golden vectors in the
reference); these closed-form checks plus the scikit-learn comparison are what holds the oracle.
"""
import numpy as np
import pytest

from oracle import oracle as O
from microaligner_amd import synthetic


# ---- Farneback ---------------------------------------------------------------------------------
def test_polyexp_constants_match_survey():
    g, xg, xxg, (ig11, ig03, ig33, ig55) = O.farneback_polyexp_constants(1, 1.7)
    np.testing.assert_allclose(g, [0.31358987, 0.37282026, 0.31358987], rtol=1e-7)
    np.testing.assert_allclose(xg, [-0.31358987, 0, 0.31358987], rtol=1e-7)
    np.testing.assert_allclose([ig11, ig03, ig33, ig55], [1.5944392544, -2.6822576782, 4.2766969327, 2.5422365360],
                               rtol=1e-7)


def test_window_kernel():
    for win, m in ((99, 49), (51, 25), (19, 9), (9, 4)):
        k = O.farneback_window_kernel(win)
        assert k.size == m + 1
        assert abs(k[0] + 2 * k[1:].sum() - 1) < 1e-6
    k = O.farneback_window_kernel(99)
    np.testing.assert_allclose([k[0], k[49]], [0.02715950, 1.05e-4], rtol=2e-3)


def test_farneback_identical_images_zero_flow_away_from_band():
    ref, _ = synthetic.make_pair(160, 150, 3)
    win, iters = 19, 3
    flow = O.calc_optical_flow_farneback(ref, ref, win, iters)
    band = iters * (win // 2) + 1
    assert np.all(flow[:-band, :-band] == 0)
    assert np.abs(flow).max() < 1.0 and np.abs(flow).max() > 0  # the bottom/right L-shaped band


def test_farneback_constant_images():
    img = np.full((64, 80), 123, np.float32)
    flow, r0, r1, m0 = O.calc_optical_flow_farneback(img, img, 9, 2, dump=True)
    assert np.abs(r0).max() < 1e-4
    assert np.abs(flow).max() < 1e-3


def test_polyexp_of_quadratic_image_returns_its_coefficients():
    H, W = 48, 56
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    a, bx, by, cxx, cyy, cxy = 5.0, 0.7, -0.4, 0.03, -0.02, 0.015
    img = (a + bx * x + by * y + cxx * x * x + cyy * y * y + cxy * x * y).astype(np.float32)
    _, r0, _, _ = O.calc_optical_flow_farneback(img, img, 9, 1, dump=True)
    inner = (slice(3, -3), slice(3, -3))
    exp = np.stack([by + 2 * cyy * y + cxy * x, bx + 2 * cxx * x + cxy * y, np.full_like(x, cyy),
                    np.full_like(x, cxx), np.full_like(x, cxy)], -1)
    np.testing.assert_allclose(r0[inner], exp[inner], atol=2e-4)


@pytest.mark.parametrize("d", [(1.5, -0.7), (-2.2, 1.1)])
def test_farneback_recovers_translation(d):
    ref, mov = synthetic.make_pair(260, 250, 11, shift=d, amp=0.0)
    flow = O.calc_optical_flow_farneback(mov, ref, 99, 3)
    inner = flow[80:-80, 80:-80].reshape(-1, 2)
    np.testing.assert_allclose(inner.mean(0), d, atol=2e-2)
    assert inner.std(0).max() < 2e-2


def test_farneback_u8_equals_f32_of_same_values():
    ref, mov = synthetic.make_pair(90, 100, 5, np.uint8)
    f8 = O.calc_optical_flow_farneback(mov, ref, 15, 2)
    f32 = O.calc_optical_flow_farneback(mov.astype(np.float32), ref.astype(np.float32), 15, 2)
    assert np.array_equal(f8, f32)


def test_fused_mode_is_close_to_unfused():
    ref, mov = synthetic.make_pair(120, 130, 8)
    a = O.calc_optical_flow_farneback(mov, ref, 31, 3)
    b = O.calc_optical_flow_farneback(mov, ref, 31, 3, fused=True)
    assert not np.array_equal(a, b)
    assert np.abs(a - b).max() < 1e-3


# ---- remap ----------------------------------------------------------------------------------------
def _grid(h, w):
    m = np.empty((h, w, 2), np.float32)
    m[..., 0] = np.arange(w)
    m[..., 1] = np.arange(h)[:, None]
    return m


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_remap_identity_and_integer_shift(dtype):
    rng = np.random.default_rng(0)
    img = (rng.random((37, 45)) * (255 if dtype != np.uint16 else 65535)).astype(dtype)
    m = _grid(37, 45)
    assert np.array_equal(O.remap(img, m), img)
    m2 = m.copy()
    m2[..., 0] += 3
    m2[..., 1] -= 2
    out = O.remap(img, m2)
    exp = np.zeros_like(img)
    exp[2:, :-3] = img[:-2, 3:]
    assert np.array_equal(out, exp)


def test_remap_u8_half_pixel_and_quantisation():
    img = np.array([[10, 21, 40, 250]], np.uint8)
    m = _grid(1, 4)
    mh = m.copy()
    mh[..., 0] += 0.5
    out = O.remap(img, mh)
    assert list(out[0]) == [(10 + 21 + 1) >> 1, (21 + 40 + 1) >> 1, (40 + 250 + 1) >> 1, (250 + 0 + 1) >> 1]
    m1 = m.copy(); m1[..., 0] += 0.01   # rounds to bucket 0
    assert np.array_equal(O.remap(img, m1), img)
    m2 = m.copy(); m2[..., 0] += 0.02   # rounds to bucket 1/32
    assert not np.array_equal(O.remap(img, m2), img)


def test_remap_tables():
    tf, ti = O.remap_tables()
    assert list(ti[0]) == [32767, 0, 0, 1]
    assert np.all(ti[1:].astype(np.int64).sum(1) == 32768)
    np.testing.assert_allclose(tf.sum(1), 1.0, rtol=0, atol=1e-7)


def test_remap_two_channel_float_and_border():
    rng = np.random.default_rng(1)
    src = rng.standard_normal((20, 22, 2)).astype(np.float32)
    m = _grid(20, 22)
    assert np.array_equal(O.remap(src, m), src)
    far = np.full((4, 4, 2), -5.0, np.float32)
    assert np.all(O.remap(src, far) == 0)


# ---- pyramids --------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(40, 50), (41, 51), (5, 7)])
def test_pyr_down_constant_and_shape(shape):
    for dtype in (np.uint8, np.uint16, np.float32):
        img = np.full(shape, 77, dtype)
        out = O.pyr_down(img)
        assert out.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
        assert np.all(out == 77)


def test_pyr_down_u8_rounding_vs_float():
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (33, 47)).astype(np.uint8)
    o8 = O.pyr_down(img)
    of = O.pyr_down(img.astype(np.float32))
    assert np.array_equal(o8, np.floor(of + 0.5).astype(np.uint8))


@pytest.mark.parametrize("src,dst", [((5, 7), (9, 13)), ((5, 7), (10, 14)), ((6, 4), (12, 7))])
def test_pyr_up_constant_and_ramp(src, dst):
    h, w = src
    c = np.full((h, w, 2), 3.5, np.float32)
    out = O.pyr_up(c, dstsize=dst[::-1])
    assert out.shape == dst + (2,)
    np.testing.assert_allclose(out, 3.5, rtol=1e-6)
    ramp = np.zeros((h, w, 2), np.float32)
    ramp[..., 0] = np.arange(w)
    up = O.pyr_up(ramp, dstsize=dst[::-1])
    j = np.arange(2, min(dst[1], 2 * w) - 3)
    np.testing.assert_allclose(up[2, j, 0], j / 2.0, atol=1e-6)


# ---- DOG ---------------------------------------------------------------------------------------------
def test_gaussian_kernel_values():
    k5, k9 = O.gaussian_kernel(41, 5), O.gaussian_kernel(41, 9)
    np.testing.assert_allclose([k5[20], k5[0]], [0.0797917, 2.68e-5], rtol=2e-3)
    np.testing.assert_allclose([k9[20], k9[0]], [0.0453551, 3.84e-3], rtol=2e-3)
    assert abs(k5.sum() - 1) < 1e-6 and abs(k9.sum() - 1) < 1e-6


def test_dog_special_cases():
    z = np.zeros((30, 30), np.float32)
    assert O.dog(z, True) is z                         # img.max() == 0 -> unchanged
    c = np.full((60, 60), 9, np.uint8)
    assert np.all(O.dog(c, True) == 0)                 # constant image -> all-zero u8
    img, _ = synthetic.make_pair(90, 80, 4)
    d = O.dog(img, True)
    assert d.dtype == np.uint8 and d.min() == 0 and d.max() == 255
    assert O.dog(img, False) is img


def test_gaussian_blur_matches_scipy_correlate():
    from scipy.ndimage import correlate1d
    img, _ = synthetic.make_pair(70, 64, 9)
    img /= 255
    k = O.gaussian_kernel(41, 5).astype(np.float64)
    exp = correlate1d(correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    np.testing.assert_allclose(O.gaussian_blur(img, 41, 5), exp, atol=2e-6)


def _round_f32(fr):
    """Correctly rounded float32 of an exact rational (round to nearest, ties to even)."""
    from fractions import Fraction
    c = np.float32(float(fr))
    cands = {float(c), float(np.nextafter(c, np.float32(np.inf))), float(np.nextafter(c, np.float32(-np.inf)))}
    def key(v):
        even = (np.float32(v).view(np.uint32) & 1) == 0
        return (abs(Fraction(v) - fr), 0 if even else 1)
    return np.float32(min(cands, key=key))


def test_gaussian_blur_fused_model_is_one_rounding_per_tap():
    """The fused rounding model (ORC_DOG_FUSED_BLUR): row filter acc = fma(k_j, x_j, acc) left to right, column filter
    acc = fma(k_j, a + b, acc) outward from the centre -- checked against exact rational arithmetic with one
    correctly rounded float32 result per tap, on a small image (pure-Python loops)."""
    from fractions import Fraction as F
    rng = np.random.default_rng(3)
    img = rng.random((7, 9)).astype(np.float32)
    ks, sigma = 5, 1.0
    k = O.gaussian_kernel(ks, sigma)
    r = ks // 2
    refl = lambda p, n: -p if p < 0 else (2 * n - 2 - p if p >= n else p)   # noqa: E731  (BORDER_REFLECT_101)
    h, w = img.shape
    rows = np.empty_like(img)
    for y in range(h):
        for x in range(w):
            acc = np.float32(k[0]) * img[y, refl(x - r, w)]
            for j in range(1, ks):
                acc = _round_f32(F(float(k[j])) * F(float(img[y, refl(x - r + j, w)])) + F(float(acc)))
            rows[y, x] = acc
    exp = np.empty_like(img)
    for y in range(h):
        for x in range(w):
            acc = np.float32(k[r]) * rows[y, x]
            for j in range(1, r + 1):
                s = rows[refl(y + j, h), x] + rows[refl(y - j, h), x]     # float32 add, rounded
                acc = _round_f32(F(float(k[r + j])) * F(float(s)) + F(float(acc)))
            exp[y, x] = acc
    got = O.gaussian_blur(img, ks, sigma, fused=True)
    assert np.array_equal(got, exp)
    assert not np.array_equal(got, O.gaussian_blur(img, ks, sigma))     # the models do differ in the last bit


def test_dog_rounding_models_differ_by_at_most_one_grey_level():
    img, _ = synthetic.make_pair(300, 280, 5)
    base = O.dog(img).astype(np.int16)
    for flags in (O.DOG_FUSED_BLUR, O.DOG_FUSED_SCALE, O.DOG_FUSED):
        d = np.abs(O.dog(img, flags=flags).astype(np.int16) - base)
        assert d.max() <= 1 and (d > 0).mean() < 0.01
    f = O.normalize_minmax_f32(img, fused=True)
    assert abs(f.min()) < 1e-7 and abs(f.max() - 1.0) < 1e-6   # fma(smin, a, -round(smin*a)) is the rounding residual, not 0
    assert np.abs(f - O.normalize_minmax_f32(img)).max() < 2e-7


# ---- NMI: pinned against the installed scikit-learn ---------------------------------------------------
def test_nmi_matches_sklearn():
    from sklearn.metrics import normalized_mutual_info_score as nmi
    rng = np.random.default_rng(5)
    for n in (1000, 50_000, 300_000):
        a = rng.integers(0, 256, n).astype(np.uint8)
        b = (a // 3 + rng.integers(0, 40, n)).astype(np.uint8)
        assert abs(O.nmi_u8(a, b) - nmi(a, b)) < 1e-12
        assert abs(O.nmi_u8(a, a) - 1.0) < 1e-12
    zeros = np.zeros(500, np.uint8)
    assert O.nmi_u8(zeros, zeros) == 1.0 == nmi(zeros, zeros)
    assert O.nmi_u8(zeros, rng.integers(0, 9, 500).astype(np.uint8)) == 0.0
    assert nmi(zeros, rng.integers(0, 9, 500)) == 0.0


def test_nmi_on_dog_images_matches_sklearn():
    from sklearn.metrics import normalized_mutual_info_score as nmi
    ref, mov = synthetic.make_pair(200, 210, 12)
    a, b = O.dog(ref, True), O.dog(mov, True)
    assert abs(O.nmi_u8(a, b) - nmi(a.ravel(), b.ravel())) < 1e-12


def test_oracle_results_do_not_depend_on_the_thread_count():
    """Image rows, NMI chunks and Farneback windows fan out over host threads (full-size parity tests and bench.py's
    cpu_baseline use every core): independent units, so the bits must not change."""
    from oracle import register_oracle as RO
    from microaligner_amd import synthetic
    ref, mov = synthetic.make_pair(333, 410, 17)
    params = dict(num_pyr_lvl=1, use_full_res_img=True, use_dog=True, tile_size=150, overlap=22)
    f1, r1 = RO.register(ref, mov, nthreads=1, **params)
    f5, r5 = RO.register(ref, mov, nthreads=5, **params)
    assert np.array_equal(f1, f5) and r1 == r5
    O.set_threads(3)
    a = O.calc_optical_flow_farneback(mov, ref, 21, 2, fused=True)
    w3 = RO.warp(mov, f1, 150, 22)
    O.set_threads(1)
    assert np.array_equal(a, O.calc_optical_flow_farneback(mov, ref, 21, 2, fused=True))
    assert np.array_equal(w3, RO.warp(mov, f1, 150, 22))


def test_preblur_taps_are_powers_of_two_so_a_fused_multiply_add_changes_nothing():
    """The 3 x 3 pre-blur inside Farneback also goes through cv::GaussianBlur (CPU-dispatched filter.simd.hpp), whose AVX2 + FMA3
    object computes  fma(a + b, k1, x * k0)  where the SSE2 baseline computes  x * k0 + (a + b) * k1.  With sigma = 0 and
    ksize = 3 the taps are the fixed [1/4, 1/2, 1/4]: both products are EXACT in float32 (a scaling by a
    power of two), so the one rounding of the fused form rounds the same real number as the final rounding of the unfused form --
    the two models cannot differ in any bit, for the row pass and for the column pass alike (short of results in the denormal
    range, |value| < 2^-126 * 4, which image data and their blurs never reach).  Hence no third rounding-model switch: the
    window blur (muladd_fused) and the dog() chain (dog_muladd_fused), whose taps are not powers of two, are the only places
    where an FMA build of OpenCV can differ.  Checked here bit for bit on a million triples over 40 binades, and on the oracle's
    own pre-blur against a float64 evaluation rounded once."""
    rng = np.random.default_rng(0)
    n = 1 << 20
    mag = np.exp2(rng.integers(-20, 20, (3, n))).astype(np.float32)
    x, a, b = (rng.standard_normal((3, n)).astype(np.float32) * mag)
    k0, k1 = np.float32(0.5), np.float32(0.25)
    s = a + b                                                   # float32: rounded once in both models
    unfused = x * k0 + s * k1                                   # three float32 operations
    fused = (x.astype(np.float64) * 0.5 + s.astype(np.float64) * 0.25).astype(np.float32)   # exact products, one rounding = fma
    assert np.array_equal(unfused, fused)
    # the products themselves are exact
    assert np.array_equal((x * k0).astype(np.float64), x.astype(np.float64) * 0.5)
    assert np.array_equal((s * k1).astype(np.float64), s.astype(np.float64) * 0.25)
