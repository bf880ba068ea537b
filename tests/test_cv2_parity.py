"""Parity against the REAL OpenCV, for the day one is at hand ("opportunistic importorskip(cv2)").

The image this repository is built in has no cv2 (no wheel, no source, no network), so every test here SKIPS there
and the oracle's OpenCV arithmetic stays "parity unpinned" (oracle/ma_oracle.c header).  Wherever
`import cv2` works these tests pin it: each primitive the hot path delegates to OpenCV is compared

  * oracle (oracle/ma_oracle.c)  vs  cv2      -- not marked gpu, runs on any host with cv2
  * HIP path (C-ABI)             vs  cv2      -- marked gpu

with the reference's own call signatures (cited per test).  Bars: bit-exact for integer results (north_star:
"warped integer output is bit-exact to cv2.remap"); flows within FLOW_TOL px of cv2.calcOpticalFlowFarneback;
float32 images within a few ulp.  The first test prints the OpenCV build facts that decide which of the known
divergence suspects apply (IPP, AVX2/FMA dispatch of filter.simd.hpp, vectorised pyrDown for float32).
"""
import os

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="OpenCV is not installed: parity with the real cv2 stays unpinned here")

from oracle import oracle as O                      # noqa: E402
from oracle import register_oracle as RO            # noqa: E402
from microaligner_amd import synthetic              # noqa: E402

FLOW_TOL = 1e-3      # px, |flow - cv2 flow| (north_star: "within a stated float32 tolerance")
F32_RTOL = 2e-6      # float32 image results (a few ulp: association order of vectorised sums)


def _pair(h, w, seed, dtype):
    return synthetic.make_pair(h, w, seed, dtype)


def _report(name, got, exp):
    diff = np.abs(got.astype(np.float64) - exp.astype(np.float64))
    print(f"[cv2 parity] {name}: exact={np.array_equal(got, exp)} max|d|={diff.max():.3g} "
          f"differing={int((diff > 0).sum())}/{diff.size}")


def test_print_opencv_build_facts():
    info = cv2.getBuildInformation()
    keep = [ln.strip() for ln in info.splitlines()
            if any(k in ln for k in ("Version control", "CPU/HW features", "Baseline", "Dispatched", "requested",
                                     "Intel IPP", "IPP", "Parallel framework", "OpenCL", "Lapack"))]
    print("\n[cv2 parity] cv2", cv2.__version__)
    for ln in keep:
        print("[cv2 parity]   ", ln)
    assert cv2.__version__.split(".")[0] == "4", "the reference pins opencv-contrib-python >=4.5,<5.0 (setup.py:40)"


# ---- the cv2 calls exactly as the reference makes them ---------------------------------------------------------
def cv_farneback(mov, ref, win, iters):
    """flow_calc.py:33-44."""
    return cv2.calcOpticalFlowFarneback(mov, ref, None, pyr_scale=0.5, levels=0, winsize=win, iterations=iters,
                                        poly_n=1, poly_sigma=1.7, flags=cv2.OPTFLOW_FARNEBACK_GAUSSIAN)


def cv_dog(img, low_sigma=5, high_sigma=9):
    """optflow_registrator.py:249-274."""
    if img.max() == 0:
        return img
    fimg = cv2.normalize(img, None, 0, 1, cv2.NORM_MINMAX, cv2.CV_32F)
    ks = (low_sigma * 4 * 2 + 1, low_sigma * 4 * 2 + 1)
    ls = cv2.GaussianBlur(fimg, ks, sigmaX=low_sigma, dst=None, sigmaY=low_sigma)
    hs = cv2.GaussianBlur(fimg, ks, sigmaX=high_sigma, dst=None, sigmaY=high_sigma)
    return cv2.normalize(hs - ls, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U)


def _random_map(h, w, sh, sw, seed):
    rng = np.random.default_rng(seed)
    m = np.empty((h, w, 2), np.float32)
    m[..., 0] = rng.uniform(-3, sw + 3, (h, w))
    m[..., 1] = rng.uniform(-3, sh + 3, (h, w))
    m[::7, ::5, 0] = np.round(m[::7, ::5, 0])            # exact pixel centres
    m[3::11, 2::9, 1] += 1.0 / 64                        # half a quantisation step: rounding of the 1/32 grid
    return m


FB_CASES = [(np.uint8, 19), (np.uint8, 99), (np.float32, 19), (np.float32, 99)]
REMAP_CASES = [(np.uint8, 1), (np.uint16, 1), (np.float32, 1), (np.float32, 2)]


# ---- oracle vs cv2 (no GPU needed) ------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,win", FB_CASES)
def test_oracle_farneback_vs_cv2(dtype, win):
    ref, mov = _pair(300, 340, 3, dtype)
    exp = cv_farneback(mov, ref, win, 3)
    got = O.calc_optical_flow_farneback(mov, ref, win, 3)
    got_fma = O.calc_optical_flow_farneback(mov, ref, win, 3, fused=True)
    _report(f"farneback {np.dtype(dtype).name} win {win} (mul+add model)", got, exp)
    _report(f"farneback {np.dtype(dtype).name} win {win} (fma model)", got_fma, exp)
    assert min(np.abs(got - exp).max(), np.abs(got_fma - exp).max()) <= FLOW_TOL


@pytest.mark.parametrize("dtype,cn", REMAP_CASES)
def test_oracle_remap_vs_cv2(dtype, cn):
    rng = np.random.default_rng(5)
    shape = (211, 263) if cn == 1 else (211, 263, cn)
    src = (rng.uniform(0, 1, shape) * (255 if dtype == np.uint8 else 60000)).astype(dtype)
    m = _random_map(190, 240, 211, 263, 6)
    exp = cv2.remap(src, m, None, cv2.INTER_LINEAR)      # warper.py:65, optflow_registrator.py:45
    got = O.remap(src, m)
    _report(f"remap {np.dtype(dtype).name} x{cn}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-3)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("shape", [(240, 320), (241, 323)])
def test_oracle_pyr_down_vs_cv2(dtype, shape):
    img, _ = _pair(*shape, 7, dtype)
    exp = cv2.pyrDown(img)                                # optflow_registrator.py:194
    got = O.pyr_down(img)
    _report(f"pyrDown {np.dtype(dtype).name} {shape}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-4)


@pytest.mark.parametrize("src_shape,dst_hw", [((120, 160), (240, 320)), ((121, 162), (241, 323)), ((121, 162), (242, 324))])
def test_oracle_pyr_up_flow_vs_cv2(src_shape, dst_hw):
    rng = np.random.default_rng(8)
    flow = rng.normal(0, 3, src_shape + (2,)).astype(np.float32)
    exp = cv2.pyrUp(flow * 2, dstsize=dst_hw[::-1])       # optflow_registrator.py:140,150,164
    got = O.pyr_up(flow * 2, dstsize=dst_hw[::-1])
    _report(f"pyrUp {src_shape}->{dst_hw}", got, exp)
    np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-5)


DOG_MODELS = {0: "SSE2 baseline (mul, add)", O.DOG_FUSED_BLUR: "fused GaussianBlur", O.DOG_FUSED_SCALE: "fused normalize",
              O.DOG_FUSED: "AVX2 + FMA3 objects (both fused)"}


def _exact_dog_models(img, exp, dog_fn):
    """Which rounding models of the dog() chain reproduce this cv2's output bit for bit."""
    return [f for f in DOG_MODELS if np.array_equal(dog_fn(img, f), exp)]


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_oracle_dog_vs_cv2(dtype):
    """GaussianBlur and convertTo live in CPU-dispatched objects: an AVX2 + FMA3 host runs them with fused
    multiply-adds, the SSE2 baseline does not (oracle/ma_oracle.c, ORC_DOG_*).  One of the four models must be exact;
    the test prints which (IPP's GaussianBlur, if this build routes there, is the remaining suspect when none is)."""
    img, _ = _pair(260, 300, 9, dtype)
    exp = cv_dog(img)
    for f, name in DOG_MODELS.items():
        _report(f"dog {np.dtype(dtype).name} [{name}]", O.dog(img, flags=f), exp)
    exact = _exact_dog_models(img, exp, lambda a, f: O.dog(a, flags=f))
    print(f"[cv2 parity] dog {np.dtype(dtype).name}: exact rounding models on this host: {[DOG_MODELS[f] for f in exact]}")
    assert exact, "no rounding model of the dog() chain reproduces this OpenCV build bit for bit"


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_oracle_normalize_u8_vs_cv2(dtype):
    img, _ = _pair(200, 230, 10, dtype)
    exp = cv2.normalize(img, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U)   # shared_modules/utils.py:94
    got = O.normalize_minmax_u8(img.astype(np.float32))
    _report(f"normalize->u8 {np.dtype(dtype).name}", got, exp)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_oracle_warp_affine_vs_cv2(dtype):
    img, _ = _pair(220, 260, 11, dtype)
    M = np.array([[0.998, -0.021, 4.3], [0.019, 1.003, -2.7]])
    exp = cv2.warpAffine(img, M, dsize=(260, 220))        # feature_registrator.py:130
    got = O.warp_affine(img, M, dsize=(260, 220))
    _report(f"warpAffine {np.dtype(dtype).name}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-3)


from oracle.cv2_backend import Cv2Prims as _Cv2Prims, register_over_cv2     # noqa: E402  (cv2 imports: checked above)


E2E = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=200, overlap=30)


def test_oracle_register_end_to_end_vs_cv2():
    ref, mov = _pair(520, 610, 12, np.float32)
    exp_flow, exp_rep, exp_warp = register_over_cv2(ref, mov, **E2E)
    flow, rep = RO.register(ref, mov, **E2E)
    _report("register() flow, oracle vs cv2", flow, exp_flow)
    assert [r[3] for r in rep] == [r[3] for r in exp_rep]
    np.testing.assert_allclose([r[1:3] for r in rep], [r[1:3] for r in exp_rep], atol=5e-3)
    assert np.abs(flow - exp_flow).max() <= 20 * FLOW_TOL   # three levels of flows composed through remaps


# ---- HIP path vs cv2 ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,win", FB_CASES)
def test_hip_farneback_vs_cv2(ctx, dtype, win):
    ref, mov = _pair(300, 340, 3, dtype)
    exp = cv_farneback(mov, ref, win, 3)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, 3).numpy()
    got_fma = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, 3, fused=True).numpy()
    _report(f"HIP farneback {np.dtype(dtype).name} win {win} (mul+add)", got, exp)
    _report(f"HIP farneback {np.dtype(dtype).name} win {win} (fma)", got_fma, exp)
    assert min(np.abs(got - exp).max(), np.abs(got_fma - exp).max()) <= FLOW_TOL


@pytest.mark.gpu
def test_hip_tiled_farneback_vs_cv2_per_window(ctx):
    """TileFlowCalc (flow_calc.py:59-98): windows through cv2 one by one, stitched, vs the batched kernels."""
    ref, mov = _pair(700, 900, 4, np.float32)
    saved = RO.O
    RO.O = _Cv2Prims
    try:
        exp = RO.tile_flow(ref, mov, 300, 40, 39, 3)
    finally:
        RO.O = saved
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 39, 3, tile=300, overlap=40).numpy()
    _report("HIP tiled farneback", got, exp)
    assert np.abs(got - exp).max() <= FLOW_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,cn", REMAP_CASES)
def test_hip_remap_vs_cv2(ctx, dtype, cn):
    rng = np.random.default_rng(5)
    shape = (211, 263) if cn == 1 else (211, 263, cn)
    src = (rng.uniform(0, 1, shape) * (255 if dtype == np.uint8 else 60000)).astype(dtype)
    m = _random_map(190, 240, 211, 263, 6)
    exp = cv2.remap(src, m, None, cv2.INTER_LINEAR)
    got = ctx.remap(ctx.asdevice(src), ctx.asdevice(m)).numpy()
    _report(f"HIP remap {np.dtype(dtype).name} x{cn}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_hip_warp_vs_cv2_per_window(ctx, dtype):
    """Warper.warp (warper.py:37-76) with cv2.remap per window vs warp_tiled_kernel: bit-exact for integers."""
    ref, mov = _pair(640, 700, 13, dtype)
    rng = np.random.default_rng(14)
    from scipy.ndimage import gaussian_filter
    flow = np.stack([gaussian_filter(rng.standard_normal((640, 700)), 8) * 60 for _ in range(2)], -1).astype(np.float32)
    saved = RO.O
    RO.O = _Cv2Prims
    try:
        exp = RO.warp(mov, flow, 250, 35)
    finally:
        RO.O = saved
    got = ctx.warp(ctx.asdevice(mov), ctx.asdevice(flow), 250, 35).numpy()
    _report(f"HIP Warper.warp {np.dtype(dtype).name}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_hip_pyramids_vs_cv2(ctx, dtype):
    img, _ = _pair(241, 323, 7, dtype)
    exp = cv2.pyrDown(img)
    got = ctx.pyr_down(ctx.asdevice(img)).numpy()
    _report(f"HIP pyrDown {np.dtype(dtype).name}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-4)
    rng = np.random.default_rng(8)
    flow = rng.normal(0, 3, (121, 162, 2)).astype(np.float32)
    for dst in ((241, 323), (242, 324)):
        exp = cv2.pyrUp(flow * 2, dstsize=dst[::-1])
        got = ctx.pyr_up_flow(ctx.asdevice(flow), dst, 2.0).numpy()
        _report(f"HIP pyrUp -> {dst}", got, exp)
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_hip_dog_and_normalize_vs_cv2(ctx, dtype):
    img, _ = _pair(260, 300, 9, dtype)
    exp = cv_dog(img)
    d = ctx.asdevice(img)
    exact = _exact_dog_models(d, exp, lambda a, f: ctx.dog_u8(a, flags=f).numpy())
    print(f"[cv2 parity] HIP dog {np.dtype(dtype).name}: exact rounding models on this host: {[DOG_MODELS[f] for f in exact]}"
          " (OptFlowRegistrator.dog_muladd_fused selects the AVX2 + FMA3 one)")
    assert exact, "no rounding model of the HIP dog() chain reproduces this OpenCV build bit for bit"
    exp = cv2.normalize(img, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U)
    got = ctx.normalize_minmax_u8(ctx.asdevice(img)).numpy()
    assert np.array_equal(got, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_hip_warp_affine_vs_cv2(ctx, dtype):
    img, _ = _pair(220, 260, 11, dtype)
    M = np.array([[0.998, -0.021, 4.3], [0.019, 1.003, -2.7]])
    exp = cv2.warpAffine(img, M, dsize=(260, 220))
    got = ctx.warp_affine_cv(ctx.asdevice(img), M, dsize=(260, 220)).numpy()
    _report(f"HIP warpAffine {np.dtype(dtype).name}", got, exp)
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=F32_RTOL, atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_hip_register_and_warp_end_to_end_vs_cv2(dtype):
    """OptFlowRegistrator.register() + Warper.warp() (optflow_registrator.py:93-173, warper.py:37-53) on the GPU vs
    the same orchestration over the real cv2: same level decisions, flow within tolerance, warped uint8 image
    within 1 LSB where the flows differ by less than the 1/32 px remap quantum."""
    from microaligner_amd import OptFlowRegistrator, Warper
    ref, mov = _pair(520, 610, 12, dtype)
    exp_flow, exp_rep, exp_warp = register_over_cv2(ref, mov, **E2E)
    reg = OptFlowRegistrator()
    reg.verbose = False
    for k, v in E2E.items():
        setattr(reg, k, v)
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    _report(f"HIP register() flow {np.dtype(dtype).name}", flow, exp_flow)
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in exp_rep]
    assert np.abs(flow - exp_flow).max() <= 20 * FLOW_TOL
    w = Warper()
    w.tile_size, w.overlap = E2E["tile_size"], E2E["overlap"]
    w.image, w.flow = mov, flow
    warped = w.warp()
    _report(f"HIP warp() {np.dtype(dtype).name}", warped, exp_warp)
    assert np.abs(warped.astype(np.float64) - exp_warp.astype(np.float64)).max() <= (2 if dtype == np.uint8 else 0.5)
