"""-m gpu: every HIP primitive against the CPU oracle on the same seeded inputs, through the C-ABI.
Integer/byte outputs and -- because the kernels follow the oracle's operation order with FMA contraction
off -- the float outputs too are required to be bit-identical (np.array_equal)."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import register_oracle as RO
from microaligner_amd import synthetic

pytestmark = pytest.mark.gpu


def pair(h, w, seed, dtype=np.float32, **kw):
    return synthetic.make_pair(h, w, seed, dtype, **kw)


# ---- Farneback ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,win,iters,dtype", [
    ((131, 157), 19, 3, np.float32),
    ((70, 64), 9, 1, np.float32),
    ((200, 300), 51, 2, np.uint8),
    ((257, 129), 99, 3, np.float32),
    ((64, 200), 15, 3, np.uint16),
    ((33, 31), 5, 2, np.float32),
    ((300, 280), 99, 3, np.uint8),
])
def test_farneback_untiled_bit_exact(ctx, shape, win, iters, dtype):
    ref, mov = pair(*shape, seed=shape[0] + win, dtype=dtype)
    exp = O.calc_optical_flow_farneback(mov, ref, win, iters)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, iters).numpy()
    assert got.shape == exp.shape and got.dtype == np.float32
    assert np.array_equal(got, exp), f"max abs diff {np.abs(got - exp).max()}"


def test_farneback_intermediates_bit_exact(ctx):
    ref, mov = pair(150, 170, 3)
    flow, r0, r1, m0 = O.calc_optical_flow_farneback(mov, ref, 21, 1, dump=True)
    gf, g0, g1, gm = [a.numpy() for a in ctx.farneback_debug(ctx.asdevice(mov), ctx.asdevice(ref), 21, 1)]
    assert np.array_equal(g0, np.moveaxis(r0, 2, 0))
    assert np.array_equal(g1, np.moveaxis(r1, 2, 0))
    assert np.array_equal(gm, np.moveaxis(m0, 2, 0))
    assert np.array_equal(gf, flow)


def test_farneback_fused_mode_bit_exact_and_close(ctx):
    ref, mov = pair(140, 150, 8)
    exp = O.calc_optical_flow_farneback(mov, ref, 31, 3, fused=True)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 31, 3, fused=True).numpy()
    assert np.array_equal(got, exp)
    plain = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 31, 3).numpy()
    assert np.abs(got - plain).max() < 1e-3


def test_farneback_huge_window_fallback(ctx):
    ref, mov = pair(40, 700, 5)
    exp = O.calc_optical_flow_farneback(mov, ref, 601, 2)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 601, 2).numpy()
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("shape,tile,ov,dtype", [
    ((420, 404), 100, 20, np.float32),
    ((437, 389), 150, 16, np.uint8),
    ((250, 610), 200, 21, np.float32),
])
def test_farneback_tiled_bit_exact(ctx, shape, tile, ov, dtype):
    ref, mov = pair(*shape, seed=tile, dtype=dtype)
    win = ov - (1 - ov % 2)
    exp = RO.tile_flow(ref, mov, tile, ov, win, 3)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, 3, tile=tile, overlap=ov).numpy()
    assert np.array_equal(got, exp), f"max abs diff {np.abs(got - exp).max()}"


@pytest.mark.parametrize("win", [3, 15, 29, 57, 77, 79, 99, 141, 143, 199, 253])
def test_farneback_every_window_size_class_bit_exact(ctx, win):
    """The window-blur kernels pick their form from the window: the streaming vertical pass (tap pairs a multiple of
    7) or the tiled one with tail taps; 3, 4 or 5 staged column chunks in the horizontal pass; beyond 253 taps the
    plain fallback kernels.  One window size from every class, untiled and tiled, against the oracle."""
    ref, mov = pair(330, 410, 50 + win)
    exp = O.calc_optical_flow_farneback(mov, ref, win, 2)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, 2).numpy()
    assert np.array_equal(got, exp)
    if win <= 99:
        exp = RO.tile_flow(ref, mov, 200, 50, win, 3)
        got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), win, 3, tile=200, overlap=50).numpy()
        assert np.array_equal(got, exp)


def test_farneback_full_size_tile_bit_exact(ctx):
    """One reference-sized window: 1200 x 1200, winsize 99, 3 iterations."""
    ref, mov = pair(1200, 1200, 42)
    exp = O.calc_optical_flow_farneback(mov, ref, 99, 3)
    got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 99, 3).numpy()
    assert np.array_equal(got, exp)
    inner = got[300:-300, 300:-300].reshape(-1, 2)
    assert np.abs(inner.mean(0) - np.array(synthetic.GLOBAL_SHIFT)).max() < 0.5


def test_farneback_batches_when_workspace_is_small(ctx):
    import ctypes as C
    from microaligner_amd import _lib as L
    ref, mov = pair(420, 404, 77)
    exp = RO.tile_flow(ref, mov, 100, 20, 19, 2)
    # one 140x140 window needs 140*192*4*20 B = 2.05 MiB of planes: 8 MiB forces batches of 3 windows
    L.check(ctx.lib.ma_ctx_set_workspace_limit(ctx.handle, 8 << 20))
    try:
        got = ctx.farneback(ctx.asdevice(mov), ctx.asdevice(ref), 19, 2, tile=100, overlap=20).numpy()
    finally:
        L.check(ctx.lib.ma_ctx_set_workspace_limit(ctx.handle, 48 << 30))
    assert np.array_equal(got, exp)


def test_farneback_argument_errors(ctx):
    ref, mov = pair(64, 64, 1)
    d = ctx.asdevice(ref)
    with pytest.raises(ValueError):
        ctx.farneback(d, d, 9, 0)
    with pytest.raises(ValueError):
        ctx.farneback(d, d, 9, 1, poly_n=3)
    with pytest.raises(ValueError):
        ctx.farneback(d, ctx.asdevice(ref[:32]), 9, 1)


# ---- remap / warp / merge --------------------------------------------------------------------------------
def rand_flow(h, w, seed, scale):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter
    f = np.stack([gaussian_filter(rng.standard_normal((h, w)), 6), gaussian_filter(rng.standard_normal((h, w)), 6)], -1)
    return (f / np.abs(f).max() * scale).astype(np.float32)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("scale", [0.0, 2.5, 40.0])
def test_warp_tiled_bit_exact(ctx, dtype, scale):
    img, _ = pair(333, 290, 9, dtype)
    flow = rand_flow(333, 290, 4, scale)
    exp = RO.warp(img, flow, 100, 12)
    got = ctx.warp(ctx.asdevice(img), ctx.asdevice(flow), 100, 12).numpy()
    assert got.dtype == img.dtype and np.array_equal(got, exp)
    if scale == 0.0:
        assert np.array_equal(got, img)


def test_warp_single_window_around_small_image(ctx):
    img, _ = pair(200, 180, 2, np.uint16)
    flow = rand_flow(200, 180, 5, 3.0)
    assert np.array_equal(ctx.warp(ctx.asdevice(img), ctx.asdevice(flow), 1000, 100).numpy(), RO.warp(img, flow, 1000, 100))


@pytest.mark.parametrize("dtype,cn", [(np.uint8, 1), (np.uint16, 1), (np.float32, 1), (np.float32, 2), (np.uint8, 2)])
def test_remap_generic_bit_exact(ctx, dtype, cn):
    rng = np.random.default_rng(3)
    sh, sw, dh, dw = 61, 83, 70, 90
    src = (rng.random((sh, sw) if cn == 1 else (sh, sw, cn)) * (65535 if dtype == np.uint16 else 255)).astype(dtype)
    m = np.empty((dh, dw, 2), np.float32)
    m[..., 0] = rng.uniform(-4, sw + 4, (dh, dw))
    m[..., 1] = rng.uniform(-4, sh + 4, (dh, dw))
    m[0, 0] = (1e12, -1e12)
    m[0, 1] = (np.nan, 3.0)
    m[0, 2] = (5.015625, 7.984375)  # exact 1/64 ties -> round-half-even buckets
    exp = O.remap(src, m)
    got = ctx.remap(ctx.asdevice(src), ctx.asdevice(m)).numpy()
    assert np.array_equal(got, exp)


def test_merge_flows_bit_exact_including_window_shortcuts(ctx):
    h, w, T, ov = 330, 310, 100, 15
    f1, f2 = rand_flow(h, w, 1, 3.0), rand_flow(h, w, 2, 2.0)
    f1[:100, :100] = 0                       # window (0,0): flow1 all zero in its centre only
    f1[100:200, 100:200] = -np.abs(f1[100:200, 100:200])
    f2[200:, 200:] = 0
    f1[0:130, 180:] = -1.0                   # padded window with only negative values -> max()==0 via padding
    exp = RO.merge_flows(f1, f2, T, ov)
    got = ctx.merge_flows(ctx.asdevice(f1), ctx.asdevice(f2), T, ov).numpy()
    assert np.array_equal(got, exp)
    z = np.zeros_like(f1)
    assert np.array_equal(ctx.merge_flows(ctx.asdevice(z), ctx.asdevice(f2), T, ov).numpy(), f2)
    assert np.array_equal(ctx.merge_flows(ctx.asdevice(f1), ctx.asdevice(z), T, ov).numpy(), RO.merge_flows(f1, z, T, ov))


@pytest.mark.parametrize("shape,T,ov", [((333, 297), 100, 15), ((260, 410), 64, 32), ((200, 200), 50, 30),
                                         ((150, 700), 128, 1), ((90, 80), 100, 20)])
def test_merge_flows_geometries(ctx, shape, T, ov):
    """Cell-based window maxima (T > 2*ov), the banded fallback (T <= 2*ov) and the untiled branch."""
    h, w = shape
    f1, f2 = rand_flow(h, w, 3, 2.5), rand_flow(h, w, 4, 1.5)
    f1[: h // 3, : w // 2] = 0
    f2[h // 2:, w // 3:] = -np.abs(f2[h // 2:, w // 3:])
    f2[h // 2:, w // 2:] = 0
    got = ctx.merge_flows(ctx.asdevice(f1), ctx.asdevice(f2), T, ov).numpy()
    assert np.array_equal(got, RO.merge_flows(f1, f2, T, ov))


@pytest.mark.parametrize("T,ov", [(100, 15), (40, 25), (0, 0)])
def test_merge_flows_nan_window_takes_the_general_branch(ctx, T, ov):
    """numpy's .max() propagates NaN, `nan == 0` is False: a window holding a NaN never takes a shortcut
    (optflow_registrator.py:38-47), even when everything else in it is <= 0."""
    h, w = 230, 260
    f1, f2 = rand_flow(h, w, 8, 2.0), rand_flow(h, w, 9, 2.0)
    f1[:120, :120] = -np.abs(f1[:120, :120])   # max would be 0 through the zero padding ...
    f1[50, 60, 1] = np.nan                      # ... but for this
    f2[100:, 100:] = 0
    f2[200, 220, 0] = np.nan
    exp = RO.merge_flows(f1, f2, T, ov) if T else RO.merge_two_flows(f1, f2)
    got = ctx.merge_flows(ctx.asdevice(f1), ctx.asdevice(f2), T, ov).numpy()
    assert np.array_equal(got, exp, equal_nan=True)
    assert np.array_equal(np.isnan(got), np.isnan(exp))


@pytest.mark.parametrize("shape,T,ov", [((330, 310), 100, 15), ((523, 777), 128, 40), ((260, 410), 64, 31), ((90, 80), 100, 20)])
def test_merge_with_cell_maxima_from_the_warps_matches_the_oracle(ctx, shape, T, ov):
    """Inside register() both flows of a merge have been read by a warp, which folds their per-cell maxima as a
    by-product (ma_warp_tiled_flowcells); the merge then takes its per-window .max() tests from those
    (ma_merge_flows_tiled_cells).  Same bits as the stand-alone merge, shortcuts and NaN windows included."""
    h, w = shape
    f1, f2 = rand_flow(h, w, 11, 3.0), rand_flow(h, w, 12, 2.0)
    f1[: h // 3, : w // 3] = 0                                    # flow1 all zero in some windows
    f1[h // 2:, : w // 4] = -np.abs(f1[h // 2:, : w // 4])        # max == 0 only through the zero padding
    f2[h // 2:, w // 2:] = 0
    f2[3, 5, 1] = np.nan
    img = np.random.default_rng(1).random((h, w)).astype(np.float32)
    d1, d2, dimg = ctx.asdevice(f1), ctx.asdevice(f2), ctx.asdevice(img)
    for f, d in ((f1, d1), (f2, d2)):
        out = ctx.warp(dimg, d, T, ov, minmax=True, flow_cells=True)
        assert d.cellkeys is not None and d.cellkeys[:2] == (T, ov)
        assert np.array_equal(out.numpy(), RO.warp(img, f, T, ov), equal_nan=True)      # the warp itself is unchanged
        lo, hi = out.minmax.numpy()
        assert (lo, hi) == (np.nanmin(out.numpy()), np.nanmax(out.numpy())) or np.isnan(out.numpy()).any()
    got = ctx.merge_flows(d1, d2, T, ov).numpy()
    exp = RO.merge_flows(f1, f2, T, ov)
    assert np.array_equal(got, exp, equal_nan=True)
    # a flow without cell maxima (or with those of another tiling) falls back to the stand-alone reduction
    assert np.array_equal(ctx.merge_flows(ctx.asdevice(f1), d2, T, ov).numpy(), exp, equal_nan=True)


def test_merge_two_flows_function(ctx):
    from microaligner_amd import merge_two_flows
    f1, f2 = rand_flow(90, 80, 6, 2.0), rand_flow(90, 80, 7, 2.0)
    assert np.array_equal(merge_two_flows(f1, f2), RO.merge_two_flows(f1, f2))


# ---- pyramids -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(200, 300), (201, 303), (5, 7), (1, 9), (100, 1), (37, 1028), (64, 2052), (9, 8),
                                   (3, 4)])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_pyr_down_bit_exact(ctx, shape, dtype):
    rng = np.random.default_rng(shape[0])
    img = (rng.random(shape) * (65535 if dtype == np.uint16 else 255)).astype(dtype)
    assert np.array_equal(ctx.pyr_down(ctx.asdevice(img)).numpy(), O.pyr_down(img))


@pytest.mark.parametrize("src,dst", [((50, 70), (100, 140)), ((50, 70), (99, 139)), ((51, 33), (102, 65)),
                                     ((2, 2), (4, 3)), ((105, 101), (210, 202)), ((50, 70), (101, 141)),
                                     ((40, 600), (80, 1200)), ((33, 515), (65, 1029)), ((1, 5), (2, 10)),
                                     ((6, 1), (12, 2)), ((3, 3), (7, 7))])
@pytest.mark.parametrize("scale", [1.0, 2.0, 4.0])
def test_pyr_up_flow_bit_exact(ctx, src, dst, scale):
    f = rand_flow(src[0], src[1], 11, 5.0)
    exp = O.pyr_up(f * np.float32(scale), dstsize=dst[::-1])
    got = ctx.pyr_up_flow(ctx.asdevice(f), dst, scale).numpy()
    assert np.array_equal(got, exp)


def test_pyr_up_rejects_incompatible_size(ctx):
    f = ctx.asdevice(rand_flow(10, 10, 1, 1.0))
    with pytest.raises(ValueError):
        ctx.pyr_up_flow(f, (25, 20), 1.0)


# ---- DOG / min-max ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("shape", [(210, 260), (45, 300), (130, 37)])
def test_dog_bit_exact(ctx, dtype, shape):
    img, _ = pair(*shape, seed=shape[1], dtype=dtype)
    exp = O.dog(img, True)
    got = ctx.dog_u8(ctx.asdevice(img)).numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, exp)


@pytest.mark.parametrize("sigmas", [(1, 2), (3, 5), (6, 9)])
def test_dog_other_sigmas_bit_exact(ctx, sigmas):
    """Kernel sizes other than the reference's 41 (generic row pass, column pass with tail taps)."""
    img, _ = pair(150, 333, seed=4)
    exp = O.dog(img, True, *sigmas)
    assert np.array_equal(ctx.dog_u8(ctx.asdevice(img), *sigmas).numpy(), exp)


@pytest.mark.parametrize("flags", [O.DOG_FUSED_BLUR, O.DOG_FUSED_SCALE, O.DOG_FUSED])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_dog_fused_rounding_models_bit_exact(ctx, dtype, flags):
    """The dog() chain in the rounding models of OpenCV's AVX2 + FMA3 objects (fused multiply-adds in the two
    GaussianBlur passes and / or in the two normalize() steps): oracle and kernels agree bit for bit in every
    combination, for the fused kernel (sigmas 5 / 9) and for the two-kernel chain (other sigmas); the models differ
    from the default by at most one grey level."""
    img, _ = pair(301, 453, seed=21 + flags, dtype=dtype)
    d = ctx.asdevice(img)
    exp = O.dog(img, True, flags=flags)
    got = ctx.dog_u8(d, flags=flags).numpy()
    assert np.array_equal(got, exp)
    assert np.abs(got.astype(np.int16) - O.dog(img, True).astype(np.int16)).max() <= 1
    for sigmas in ((3, 5), (6, 9)):
        assert np.array_equal(ctx.dog_u8(d, *sigmas, flags=flags).numpy(), O.dog(img, True, *sigmas, flags=flags))
    with pytest.raises(ValueError):
        ctx.dog_u8(d, flags=4)


@pytest.mark.parametrize("shape", [(70, 1030), (33, 1024), (9, 517)])
def test_dog_wide_rows_bit_exact(ctx, shape):
    """Several 256-column blocks per row, widths that are and are not multiples of 4."""
    img, _ = pair(*shape, seed=shape[1])
    assert np.array_equal(ctx.dog_u8(ctx.asdevice(img)).numpy(), O.dog(img, True))


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
def test_producers_leave_min_max_for_dog(ctx, dtype):
    """warp / pyr_down with minmax=True reduce their own output; dog_u8 then skips its pass and gives the same bits."""
    img, _ = pair(333, 517, seed=8, dtype=dtype)
    flow = rand_flow(333, 517, 9, 4.0)
    d = ctx.asdevice(img)
    w = ctx.warp(d, ctx.asdevice(flow), 100, 20, minmax=True)
    wn = w.numpy()
    assert np.array_equal(wn, ctx.warp(d, ctx.asdevice(flow), 100, 20).numpy())
    assert tuple(w.minmax.numpy()) == (np.float32(wn.min()), np.float32(wn.max()))
    assert np.array_equal(ctx.dog_u8(w).numpy(), O.dog(wn, True))
    p = ctx.pyr_down(d, minmax=True)
    pn = p.numpy()
    assert np.array_equal(pn, O.pyr_down(img))
    assert tuple(p.minmax.numpy()) == (np.float32(pn.min()), np.float32(pn.max()))
    assert np.array_equal(ctx.dog_u8(p).numpy(), O.dog(pn, True))
    tiny = ctx.pyr_down(ctx.asdevice(img[:3, :5].copy()), minmax=True)      # per-pixel path only
    assert tuple(tiny.minmax.numpy()) == (np.float32(tiny.numpy().min()), np.float32(tiny.numpy().max()))
    z = ctx.warp(ctx.asdevice(np.zeros((50, 60), dtype)), ctx.zeros((50, 60, 2), np.float32), 0, 0, minmax=True)
    out, zero = ctx.dog_u8(z, report_zero=True)
    assert zero and not out.numpy().any()


def test_dog_special_cases_and_minmax(ctx):
    from microaligner_amd import OptFlowRegistrator
    reg = OptFlowRegistrator()
    z = np.zeros((40, 40), np.float32)
    assert reg.dog(z, True) is z and reg.dog(z, False) is z
    c = np.full((64, 64), 9, np.uint8)
    assert np.all(reg.dog(c, True) == 0)
    img, _ = pair(99, 77, 1)
    assert ctx.minmax(ctx.asdevice(img)) == (float(img.min()), float(img.max()))
    assert np.array_equal(reg.dog(img, True), O.dog(img, True))


def test_max_project_and_normalize_u8(ctx):
    rng = np.random.default_rng(0)
    stack = rng.integers(0, 60000, (5, 120, 130)).astype(np.uint16)
    mp = ctx.max_project(ctx.asdevice(stack))
    assert np.array_equal(mp.numpy(), stack.max(0))
    got = ctx.normalize_minmax_u8(mp).numpy()
    exp = O.normalize_minmax_u8(stack.max(0).astype(np.float32))
    assert np.array_equal(got, exp)


# ---- NMI ------------------------------------------------------------------------------------------------------
def test_nmi_matches_oracle_and_sklearn(ctx):
    from sklearn.metrics import normalized_mutual_info_score as nmi
    ref, mov = pair(300, 310, 12)
    a, b = O.dog(ref, True), O.dog(mov, True)
    got = ctx.nmi_scores(ctx.asdevice(a), ctx.asdevice(b), 0)
    assert got.shape == (1,)
    assert abs(got[0] - O.nmi_u8(a, b)) < 1e-12
    assert abs(got[0] - nmi(a.ravel(), b.ravel())) < 1e-12
    chunk = 100 * 100
    got = ctx.nmi_scores(ctx.asdevice(a), ctx.asdevice(b), chunk)
    fa, fb = a.ravel(), b.ravel()
    exp = [O.nmi_u8(fa[i:i + chunk], fb[i:i + chunk]) for i in range(0, fa.size, chunk)]
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-12)


def test_nmi_special_cases(ctx):
    z = np.zeros((50, 60), np.uint8)
    r = np.random.default_rng(1).integers(0, 7, (50, 60)).astype(np.uint8)
    dz, dr = ctx.asdevice(z), ctx.asdevice(r)
    assert ctx.nmi_scores(dz, dz)[0] == 1.0
    assert ctx.nmi_scores(dz, dr)[0] == 0.0
    assert abs(ctx.nmi_scores(dr, dr)[0] - 1.0) < 1e-12


def test_gate_accepts_the_unchanged_float_image_the_reference_dog_returns(ctx):
    """A float image whose maximum is 0 but which is not all zero comes out of the reference's dog() unchanged
    (optflow_registrator.py:256-257) and scikit-learn then labels its raw values: the gate takes that input too."""
    import warnings
    from sklearn.metrics import normalized_mutual_info_score as nmi
    from microaligner_amd import OptFlowRegistrator
    from microaligner_amd.shared_modules.similarity_scoring import check_if_higher_similarity, mi_tiled
    rng = np.random.default_rng(4)
    neg = -np.round(rng.random((90, 70)) * 5).astype(np.float32)       # max() == 0, values in {-5 .. 0}
    other = -np.round(rng.random((90, 70)) * 3).astype(np.float32)
    reg = OptFlowRegistrator()
    assert reg.dog(neg, True) is neg
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert abs(mi_tiled(neg, other, 1000) - nmi(neg.ravel(), other.ravel())) < 1e-14
        exp = np.mean([nmi(neg.ravel()[s:s + 400], other.ravel()[s:s + 400]) for s in range(0, neg.size, 400)])
    assert abs(mi_tiled(neg, other, 20) - exp) < 1e-14
    assert check_if_higher_similarity(neg, neg, other, 1000, verbose=False) == [True]


def test_mi_tiled_host_function(ctx):
    from microaligner_amd.shared_modules.similarity_scoring import mi_tiled, check_if_higher_similarity
    ref, mov = pair(260, 250, 3)
    a, b = O.dog(ref, True), O.dog(mov, True)
    assert abs(mi_tiled(a, b, 100) - RO.mi_tiled(a, b, 100)) < 1e-12      # tiled: 7 chunks
    assert abs(mi_tiled(a, b, 1000) - RO.mi_tiled(a, b, 1000)) < 1e-12    # whole image
    assert check_if_higher_similarity(a, a, b, 100, verbose=False) == [True]
    z = np.zeros((260, 250), np.float32)
    assert mi_tiled(z, z, 1000) == 1.0


@pytest.mark.parametrize("shape,tile", [((260, 250), 100), ((260, 250), 1000), ((2300, 2100), 1000)])
def test_both_halves_of_the_gate_in_one_call(ctx, shape, tile):
    """mutual_information_test on three device dog() outputs goes through ma_nmi_u8_pair (shared reference labels, one pair
    of launches, one synchronisation): the very scores of two separate mi_tiled calls, which equal the oracle's."""
    from microaligner_amd.shared_modules.similarity_scoring import mi_tiled, mutual_information_test
    ref, mov = pair(*shape, 5)
    other, _ = pair(*shape, 6)
    a, b0, b1 = (ctx.asdevice(O.dog(v, True)) for v in (ref, mov, other))
    after, before = mutual_information_test(a, b0, b1, tile)
    assert after == mi_tiled(a, b0, tile) and before == mi_tiled(a, b1, tile)
    assert abs(after - RO.mi_tiled(a.numpy(), b0.numpy(), tile)) < 1e-12
    assert abs(before - RO.mi_tiled(a.numpy(), b1.numpy(), tile)) < 1e-12
