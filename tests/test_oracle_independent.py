"""The C oracle against INDEPENDENT statements of the same operations, made from third-party code that IS in this image
(scipy.ndimage) and from the published algorithm (Farneback, "Two-Frame Motion Estimation Based on Polynomial Expansion",
SCIA 2003) in float64 numpy -- not from OpenCV's loops and sharing no code with oracle/ma_oracle.c.

What this pins: CONVENTIONS -- map channel order and sign, pixel-centre coordinates, the 1/32-px quantisation of cv2.remap,
border modes (constant 0 for remap, reflect-101 for pyrDown), the [1 4 6 4 1] kernel and its x4 gain in pyrUp, which image
is `prev` and which way the flow points, the polynomial basis and the Gaussian window of calcOpticalFlowFarneback.
What it does NOT pin: last bits (OpenCV's summation order, float32 intermediates, SSE2 vs FMA) -- that needs a real cv2
(tests/test_cv2_golden.py, tests/test_cv2_parity.py), which this image and the GPU pool do not have.  Tolerances are stated
per test.  Reference call sites: optflow_reg/flow_calc.py:33-44, optflow_reg/warper.py:56-66,
optflow_reg/optflow_registrator.py:45,140-214."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import oracle as O


def _texture(h, w, seed, amp=100.0):
    rng = np.random.default_rng(seed)
    img = ndi.gaussian_filter(rng.standard_normal((h, w)), 2.0)
    img = (img - img.min()) / (img.max() - img.min())
    return (img * amp).astype(np.float32)


# ---- cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) ---------------------------------------------------------------------
def _quantised_map(h, w, seed, spread):
    """Random map whose coordinates are exact multiples of 1/32 px (so that cv2's INTER_BITS = 5 quantisation is the
    identity and plain bilinear interpolation is the same function), reaching beyond every border."""
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    mx = np.round((gx + rng.uniform(-spread, spread, (h, w))) * 32) / 32
    my = np.round((gy + rng.uniform(-spread, spread, (h, w))) * 32) / 32
    return np.stack([mx, my], -1).astype(np.float32)     # channel 0 = x (column), channel 1 = y (row)


def _scipy_bilinear(src, m):
    # 'grid-constant': samples between the last pixel and the constant region are interpolated, as BORDER_CONSTANT does
    return ndi.map_coordinates(src.astype(np.float64), [m[..., 1].astype(np.float64), m[..., 0].astype(np.float64)],
                               order=1, mode="grid-constant", cval=0.0)


def test_remap_f32_is_bilinear_interpolation_at_the_map_coordinates():
    src = _texture(83, 97, 1)
    m = _quantised_map(83, 97, 2, 9.0)
    got = O.remap(src, m)
    exp = _scipy_bilinear(src, m)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, exp, rtol=1e-5, atol=1e-5 * 100)
    # sign and channel order are not interchangeable on this input: the transposed / negated readings are far off
    assert np.abs(got - _scipy_bilinear(src, m[..., ::-1])).max() > 1.0


def test_remap_quantises_coordinates_to_a_32nd_of_a_pixel():
    """cv2.remap rounds map coordinates to 1/32 px (INTER_BITS = 5) before interpolating: an arbitrary map gives what
    plain bilinear interpolation gives at the ROUNDED coordinates (round-half-even, cvRound)."""
    src = _texture(60, 70, 3)
    rng = np.random.default_rng(4)
    gx, gy = np.meshgrid(np.arange(70, dtype=np.float32), np.arange(60, dtype=np.float32))
    m = np.stack([gx + rng.uniform(-3, 3, gx.shape).astype(np.float32), gy + rng.uniform(-3, 3, gx.shape).astype(np.float32)], -1)
    mq = (np.rint(m.astype(np.float64) * 32) / 32).astype(np.float32)
    np.testing.assert_allclose(O.remap(src, m), _scipy_bilinear(src, mq), rtol=1e-5, atol=1e-3)
    assert np.abs(O.remap(src, m) - _scipy_bilinear(src, m)).max() > 1e-3      # the unquantised reading differs


@pytest.mark.parametrize("dtype,top", [(np.uint8, 255), (np.uint16, 65535)])
def test_remap_integer_images_round_the_bilinear_value(dtype, top):
    rng = np.random.default_rng(5)
    src = rng.integers(0, top + 1, (64, 80)).astype(dtype)
    m = _quantised_map(64, 80, 6, 5.0)
    got = O.remap(src, m).astype(np.float64)
    exp = _scipy_bilinear(src, m)
    # uint8: OpenCV's fixed-point weights (15 bits) and (acc + 2^14) >> 15; uint16: float weights, saturating round
    assert np.abs(got - exp).max() <= (1.0 if dtype == np.uint8 else 0.5 + 1e-2)


def test_remap_two_channel_flow_composition():
    """merge_two_flows (optflow_registrator.py:37-47): remap(flow2, -flow1) samples flow2 at the NEGATED flow1 read as
    absolute coordinates."""
    f2 = np.stack([_texture(50, 60, 7, 4.0), _texture(50, 60, 8, 4.0)], -1)
    m = _quantised_map(50, 60, 9, 2.0)
    got = O.remap(f2, m)
    for c in range(2):
        np.testing.assert_allclose(got[..., c], _scipy_bilinear(f2[..., c], m), rtol=1e-5, atol=1e-5)


# ---- cv2.pyrDown / cv2.pyrUp ----------------------------------------------------------------------------------------
K5 = np.array([1, 4, 6, 4, 1], np.float64) / 16


@pytest.mark.parametrize("shape", [(64, 80), (65, 81), (33, 40)])
def test_pyr_down_is_the_5_tap_binomial_on_a_reflect_101_border_then_every_second_pixel(shape):
    img = _texture(*shape, 11)
    full = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), K5, axis=0, mode="mirror"), K5, axis=1, mode="mirror")
    exp = full[::2, ::2]
    got = O.pyr_down(img)
    assert got.shape == ((shape[0] + 1) // 2, (shape[1] + 1) // 2)
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6 * 100)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16])
def test_pyr_down_integer_images_round_to_nearest(dtype):
    rng = np.random.default_rng(12)
    img = rng.integers(0, 256 if dtype == np.uint8 else 65536, (50, 62)).astype(dtype)
    full = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), K5, axis=0, mode="mirror"), K5, axis=1, mode="mirror")
    got = O.pyr_down(img).astype(np.float64)
    assert np.abs(got - full[::2, ::2]).max() <= 0.5 + 1e-9            # (sum + 128) >> 8 is round-half-up of sum / 256


@pytest.mark.parametrize("src,dst", [((40, 50), (80, 100)), ((40, 50), (79, 99))])
def test_pyr_up_is_zero_insertion_times_four_through_the_same_kernel(src, dst):
    """pyrUp(flow * k, dstsize) at the optflow_registrator.py:140,150,164,169,212,214 sites: rows / columns of zeros
    inserted, the binomial kernel with gain 2 per axis.  Compared away from the border (OpenCV's border rule there is its
    own: pinned by the closed-form cases of tests/test_oracle_kat.py, not by scipy)."""
    flow = np.stack([_texture(*src, 13, 5.0), _texture(*src, 14, 5.0)], -1)
    got = O.pyr_up(flow, dstsize=dst[::-1])
    assert got.shape == dst + (2,)
    for c in range(2):
        up = np.zeros((2 * src[0], 2 * src[1]))
        up[::2, ::2] = flow[..., c]
        exp = ndi.correlate1d(ndi.correlate1d(up, 2 * K5, axis=0, mode="constant"), 2 * K5, axis=1, mode="constant")
        h, w = min(dst[0], up.shape[0]) - 4, min(dst[1], up.shape[1]) - 4
        np.testing.assert_allclose(got[4:h, 4:w, c], exp[4:h, 4:w], rtol=1e-6, atol=1e-5)


# ---- cv2.calcOpticalFlowFarneback(levels=0, poly_n=1, poly_sigma=1.7, OPTFLOW_FARNEBACK_GAUSSIAN) ------------------
def _poly_expansion(img, n=1, sigma=1.7):
    """Farneback 2003, section 2: per pixel the weighted least-squares fit  f(x) ~ x'Ax + b'x + c  over a (2n+1)^2
    neighbourhood with a Gaussian applicability.  Returns c, bx, by, axx, ayy, axy (axy = coefficient of x*y)."""
    xs = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-xs ** 2 / (2 * sigma ** 2))
    X, Y = np.meshgrid(xs, xs)                                 # X varies along columns
    B = np.stack([np.ones_like(X), X, Y, X * X, Y * Y, X * Y], -1).reshape(-1, 6)
    Wa = np.outer(g, g).reshape(-1)
    proj = np.linalg.inv(B.T @ (Wa[:, None] * B)) @ (B.T * Wa)  # 6 x (2n+1)^2
    return [ndi.correlate(img, proj[k].reshape(2 * n + 1, 2 * n + 1), mode="nearest") for k in range(6)]


def _farneback_float64(prev, nxt, winsize, iterations, det_eps=0.0):
    """Displacement estimation of the paper's sections 4 - 5 (eqs. 7 - 11 with the a-priori displacement of section 5,
    iterated), float64 throughout.  OpenCV specifics that are PARAMETERS of the call, not of the paper, taken from the
    call site: 3 x 3 binomial pre-smoothing of both images at pyramid scale 1, a Gaussian window of sigma = 0.3 * (winsize
    // 2), the second expansion sampled bilinearly at x + d.  Not modelled: OpenCV's border attenuation (5 px) -- compare away
    from borders.  det_eps: OpenCV adds 1e-3 to the determinant of the 2 x 2 system (a regulariser the paper does not have);
    0 is the paper."""
    pre = lambda im: ndi.correlate1d(ndi.correlate1d(im.astype(np.float64), [0.25, 0.5, 0.25], axis=0, mode="mirror"),
                                     [0.25, 0.5, 0.25], axis=1, mode="mirror")
    c0, bx0, by0, axx0, ayy0, axy0 = _poly_expansion(pre(prev))
    r1 = _poly_expansion(pre(nxt))
    h, w = prev.shape
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    m = winsize // 2
    k = np.exp(-np.arange(-m, m + 1) ** 2 / (2 * (m * 0.3) ** 2))
    k /= k.sum()
    blur = lambda a: ndi.correlate1d(ndi.correlate1d(a, k, axis=0, mode="nearest"), k, axis=1, mode="nearest")
    dx, dy = np.zeros((h, w)), np.zeros((h, w))
    for _ in range(iterations):
        s = [ndi.map_coordinates(p, [gy + dy, gx + dx], order=1, mode="nearest") for p in r1]
        _, bx1, by1, axx1, ayy1, axy1 = s
        a11, a22, a12 = (axx0 + axx1) / 2, (ayy0 + ayy1) / 2, (axy0 + axy1) / 4       # A = (A1 + A2) / 2, A12 = axy / 2
        dbx = -0.5 * (bx1 - bx0) + a11 * dx + a12 * dy                                # eq. 10 with the a-priori d
        dby = -0.5 * (by1 - by0) + a12 * dx + a22 * dy
        g11, g12, g22 = blur(a11 * a11 + a12 * a12), blur(a12 * (a11 + a22)), blur(a12 * a12 + a22 * a22)
        h1, h2 = blur(a11 * dbx + a12 * dby), blur(a12 * dbx + a22 * dby)
        det = g11 * g22 - g12 * g12 + det_eps
        dx, dy = (g22 * h1 - g12 * h2) / det, (g11 * h2 - g12 * h1) / det             # eq. 9: d = (sum w A'A)^-1 sum w A'db
    return np.stack([dx, dy], -1)


@pytest.mark.parametrize("winsize,iterations,shift,hw", [(15, 3, (1.3, -0.7), (110, 128)), (21, 2, (-0.6, 0.9), (110, 128)),
                                                         (9, 3, (0.4, 0.3), (110, 128)),
                                                         (99, 3, (2.2, -1.4), (400, 430))])   # the reference's window and iterations
def test_farneback_equals_the_papers_normal_equations_in_float64(winsize, iterations, shift, hw):
    h, w = hw
    base = _texture(h + 20, w + 20, 20 + winsize, 200.0)
    prev = base[10:-10, 10:-10].copy()
    # next(y, x) = prev(y - sy, x - sx): content moves by +shift, the flow prev -> next is +shift
    nxt = ndi.shift(base.astype(np.float64), (shift[1], shift[0]), order=3, mode="nearest")[10:-10, 10:-10].astype(np.float32)
    got = O.calc_optical_flow_farneback(prev, nxt, winsize, iterations)
    # OpenCV attenuates the matrices within 5 px of the border (not in the paper); every iteration carries that band one
    # window radius further in, the expansion and the pre-smoothing 3 px
    b = 5 + iterations * (winsize // 2) + 3
    inner = lambda a: a[b:-b, b:-b]
    # (a) the paper as it stands: the only difference left is OpenCV's 1e-3 on the determinant, which matters where a
    #     small window sees little texture
    d = np.abs(inner(got) - inner(_farneback_float64(prev, nxt, winsize, iterations)))
    assert np.median(d) <= 2e-4 and d.max() <= (1e-3 if winsize >= 15 else 5e-3), f"paper: max {d.max():.2e} px"
    # (b) with that one documented constant the float32 oracle IS the float64 normal equations (measured: 3e-6 px)
    d = np.abs(inner(got) - inner(_farneback_float64(prev, nxt, winsize, iterations, det_eps=1e-3)))
    assert d.max() <= 2e-5, f"max |oracle - float64 normal equations| = {d.max():.2e} px"
    # and the estimate is the displacement that was applied: direction, channel order (x first), magnitude
    med = np.median(got[b:-b, b:-b].reshape(-1, 2), axis=0)
    assert abs(med[0] - shift[0]) < 0.1 and abs(med[1] - shift[1]) < 0.1


def test_farneback_window_and_expansion_constants_follow_from_the_definitions():
    """The tap table and the inverse Gram matrix entries the kernels are fed (oracle.farneback_window_kernel /
    farneback_polyexp_constants) from their definitions in float64."""
    k = O.farneback_window_kernel(99)
    m = 49
    ref = np.exp(-np.arange(0, m + 1) ** 2 / (2 * (m * 0.3) ** 2))
    ref /= ref[0] + 2 * ref[1:].sum()
    np.testing.assert_allclose(np.asarray(k)[:m + 1], ref, rtol=2e-6)


# ---- Warper.warp (optflow_reg/warper.py:37-76) ------------------------------------------------------------------------
def test_tiled_warp_is_backward_sampling_at_x_minus_flow():
    """The reference warps window by window with map = grid - flow.  While the flow stays below the overlap the windows
    are invisible: the result is the image sampled at (x - fx, y - fy) -- constant 0 beyond the IMAGE border only."""
    from oracle import register_oracle as RO
    h, w, tile, ov = 150, 170, 60, 12
    img = _texture(h, w, 31)
    rng = np.random.default_rng(32)
    flow = np.stack([ndi.gaussian_filter(rng.standard_normal((h, w)), 6) * 40, ndi.gaussian_filter(rng.standard_normal((h, w)), 6) * 40], -1)
    flow = (np.round(np.clip(flow, -ov + 2, ov - 2) * 32) / 32).astype(np.float32)
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    exp = _scipy_bilinear(img, np.stack([gx - flow[..., 0], gy - flow[..., 1]], -1))
    np.testing.assert_allclose(RO.warp(img, flow, tile, ov), exp, rtol=1e-5, atol=1e-3)


# ---- OptFlowRegistrator.dog (optflow_registrator.py:249-274) ----------------------------------------------------------
@pytest.mark.parametrize("dtype,top", [(np.uint8, 255), (np.uint16, 60000), (np.float32, 1.0)])
def test_dog_chain_in_float64(dtype, top):
    """normalize to [0, 1] -> GaussianBlur(ksize = 8 * low_sigma + 1 for BOTH sigmas, reflect-101) with sigma 5 and 9 ->
    high - low -> normalize to [0, 255] uint8, restated with scipy in float64 (kernel taps from the definition, truncated at the
    41-tap window and renormalised): the oracle's uint8 image within one grey level, equal at > 97 % of the pixels."""
    img = (_texture(180, 210, 41, 1.0) * top).astype(dtype)
    f = img.astype(np.float64)
    f = (f - f.min()) / (f.max() - f.min())

    def blur(a, sigma, ksize=41):
        x = np.arange(ksize) - ksize // 2
        k = np.exp(-x ** 2 / (2.0 * sigma ** 2))
        k /= k.sum()
        return ndi.correlate1d(ndi.correlate1d(a, k, axis=0, mode="mirror"), k, axis=1, mode="mirror")
    d = blur(f, 9) - blur(f, 5)
    exp = np.rint((d - d.min()) / (d.max() - d.min()) * 255)
    got = O.dog(img, True).astype(np.float64)
    assert O.dog(img, True).dtype == np.uint8
    assert np.abs(got - exp).max() <= 1 and (got == exp).mean() > 0.97
