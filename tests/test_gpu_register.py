"""-m gpu: the whole hot path -- OptFlowRegistrator.register() + Warper.warp() on the device -- against
(a) fixtures produced by the reference's own orchestration (tests/golden), (b) the oracle orchestration on
fresh inputs, and (c) size-independent properties at BASELINE sizes."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import register_oracle as RO
from microaligner_amd import OptFlowRegistrator, Warper, synthetic
from microaligner_amd.device import DeviceArray

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLDEN, "index.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_reg(params):
    reg = OptFlowRegistrator()
    reg.verbose = False
    for k, v in params.items():
        setattr(reg, k, v)
    return reg


@pytest.mark.parametrize("name", sorted(INDEX))
def test_register_and_warp_reproduce_reference_fixtures(name):
    case = INDEX[name]
    H, W = case["shape"]
    make = synthetic.make_unrelated_pair if case["unrelated"] else synthetic.make_pair
    ref, mov = make(H, W, case["seed"], case["dtype"])
    reg = make_reg(case["params"])
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    s5 = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert isinstance(flow, np.ndarray) and flow.dtype == np.float32 and list(flow.shape) == case["flow_shape"]
    assert [r.factor for r in reg.level_reports] == case["factors"]
    assert [r.accepted for r in reg.level_reports] == case["accepted"]
    np.testing.assert_allclose([(r.mi_after, r.mi_before) for r in reg.level_reports], case["mi"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(flow[::5, ::5], s5["flow_s5"])
    assert sha(flow) == case["flow_sha256"]

    w = Warper()
    w.tile_size = case["params"].get("tile_size", 1000)
    w.overlap = case["params"].get("overlap", 100)
    w.image, w.flow = mov, flow
    warped = w.warp()
    assert len(w.image) == 0 and len(w.flow) == 0  # inputs are consumed like the reference (Q6)
    assert warped.dtype == mov.dtype and sha(warped) == case["warped_sha256"]
    mov16 = synthetic._cast(synthetic.make_pair(H, W, case["seed"], np.float32)[1], np.uint16)
    w.image, w.flow = mov16, flow
    assert sha(w.warp()) == case["warped_u16_sha256"]


@pytest.mark.parametrize("shape,dtype,params", [
    ((512, 512), np.float32, dict()),                                                              # BASELINE cfg1
    ((640, 520), np.uint8, dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=128, overlap=24)),
    ((500, 700), np.uint16, dict(num_pyr_lvl=1, use_full_res_img=True, tile_size=200, overlap=31)),
])
def test_register_matches_oracle_orchestration(shape, dtype, params):
    ref, mov = synthetic.make_pair(*shape, seed=21, dtype=dtype)
    exp, reports = RO.register(ref, mov, **params)
    reg = make_reg(params)
    reg.ref_img, reg.mov_img = ref, mov
    got = reg.register()
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports]
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("dog_fused,fb_fused", [(True, False), (True, True)])
def test_register_in_the_fused_rounding_models_matches_the_oracle(dog_fused, fb_fused):
    """register() with the dog() chain (and the Farneback window blur) in the fused multiply-add models -- what an
    OpenCV build dispatching to its AVX2 + FMA3 objects (and one whose v_muladd is an FMA) computes: same gate
    decisions, same MI scores, same flow as the oracle orchestration in the same model."""
    from oracle import oracle as O
    params = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=160, overlap=30)
    ref, mov = synthetic.make_pair(600, 520, seed=33)
    exp, reports = RO.register(ref, mov, fused=fb_fused, dog_flags=O.DOG_FUSED if dog_fused else 0, **params)
    reg = make_reg(dict(params, dog_muladd_fused=dog_fused, muladd_fused=fb_fused))
    reg.ref_img, reg.mov_img = ref, mov
    got = reg.register()
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports]
    np.testing.assert_allclose([(r.mi_after, r.mi_before) for r in reg.level_reports], [r[1:3] for r in reports],
                               rtol=0, atol=1e-12)
    assert np.array_equal(got, exp)
    assert np.array_equal(reg.dog(ref, True), O.dog(ref, True, flags=O.DOG_FUSED if dog_fused else 0))


@pytest.mark.parametrize("unrelated,dtype,params", [
    (False, np.float32, dict(num_pyr_lvl=3, use_full_res_img=True, use_dog=True, tile_size=150, overlap=30)),
    (False, np.uint8, dict(num_pyr_lvl=2, use_full_res_img=False, tile_size=200, overlap=40)),
    (True, np.float32, dict(num_pyr_lvl=3, use_full_res_img=True, tile_size=150, overlap=30)),      # rejected levels
    (True, np.uint16, dict(num_pyr_lvl=2, use_full_res_img=False, use_dog=True, tile_size=120, overlap=25)),
    (False, np.float32, dict(num_pyr_lvl=0, use_full_res_img=True, tile_size=300, overlap=50)),     # a single level, in place
    (True, np.float32, dict(num_pyr_lvl=0, use_full_res_img=True, tile_size=300, overlap=50)),
    (False, np.float32, dict(num_pyr_lvl=1, use_full_res_img=False, tile_size=1000, overlap=100)),  # untiled, one pyrUp
])
def test_c_level_loop_equals_the_python_level_loop(unrelated, dtype, params):
    """ma_optflow_register (the level loop, gate and bookkeeping in C++ behind one entry point) against the same loop
    stated in Python over the primitive entry points: flows, per-level decisions and MI scores identical, on accepted
    and rejected levels, every dtype, with and without the full-resolution level."""
    make = synthetic.make_unrelated_pair if unrelated else synthetic.make_pair
    ref, mov = make(840, 700, 17, dtype)
    out = {}
    for engine in ("c", "python"):
        reg = make_reg(dict(params, engine=engine))
        reg.ref_img, reg.mov_img = ref, mov
        out[engine] = (reg.register(), [(r.factor, r.shape, r.mi_after, r.mi_before, r.accepted) for r in reg.level_reports])
    assert out["c"][1] == out["python"][1]
    assert out["c"][0].shape == ref.shape + (2,) and np.array_equal(out["c"][0], out["python"][0])
    exp, reports = RO.register(ref, mov, **params)
    assert np.array_equal(out["c"][0], exp) and [r[4] for r in out["c"][1]] == [r[3] for r in reports]


def test_float_image_with_zero_max_follows_the_reference(ctx):
    """A float image whose max() is exactly 0 without being all zero (here: negated images; and every warp of such an
    image, whose constant border is 0) comes out of the reference's dog() UNCHANGED (optflow_registrator.py:256-257): the
    reference goes on with the raw float image as Farneback input and as scikit-learn labels.  The C entry point reports
    the case (ma_last_error: "max() == 0"), OptFlowRegistrator repeats the call with its Python level loop, which
    follows the reference there: decisions, MI scores and flow equal the oracle orchestration (which labels through
    scikit-learn itself)."""
    from microaligner_amd import _lib as L
    ref0, mov0 = synthetic.make_pair(520, 610, seed=5)
    ref, mov = -(ref0 - ref0.min()), -(mov0 - mov0.min())
    assert ref.max() == 0 and mov.max() == 0 and ref.min() < 0 and mov.min() < 0
    params = dict(num_pyr_lvl=1, use_full_res_img=True, use_dog=True, tile_size=200, overlap=30)
    exp, reports = RO.register(ref, mov, **params)
    with pytest.raises(ValueError, match=r"max\(\) == 0"):
        ctx.optflow_register(ctx.asdevice(ref), ctx.asdevice(mov), **params)
    for engine in ("c", "python"):
        reg = make_reg(dict(params, engine=engine))
        reg.ref_img, reg.mov_img = ref, mov
        got = reg.register()
        assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports]
        np.testing.assert_allclose([(r.mi_after, r.mi_before) for r in reg.level_reports], [r[1:3] for r in reports],
                                   rtol=0, atol=1e-12)
        assert np.array_equal(got, exp)
    # the flag is per call: an ordinary pair right after it goes through the C entry point again
    flow, rep = ctx.optflow_register(ctx.asdevice(ref0), ctx.asdevice(mov0), **params)
    assert np.array_equal(flow.numpy(), RO.register(ref0, mov0, **params)[0])
    # an all-zero float image is not that case (it IS the all-zero label image): no error
    zero = np.zeros_like(ref0)
    flow, rep = ctx.optflow_register(ctx.asdevice(ref0), ctx.asdevice(zero), **params)
    assert np.array_equal(flow.numpy(), RO.register(ref0, zero, **params)[0])


@pytest.mark.parametrize("dt_ref,dt_mov,use_dog", [(np.uint8, np.uint16, False), (np.float32, np.uint8, False),
                                                   (np.uint16, np.float32, True)])
def test_reference_and_moving_image_of_different_dtypes(ctx, dt_ref, dt_mov, use_dog):
    """cv2.calcOpticalFlowFarneback converts each input to float32 on its own, so the reference accepts e.g. a uint8
    reference image with a uint16 moving image (flow_calc.py:33-44); every other step handles one image at a time in
    its own dtype.  register(), TileFlowCalc and farneback() take such pairs: results equal the oracle's."""
    from microaligner_amd import TileFlowCalc, farneback
    from oracle import oracle as O
    ref, _ = synthetic.make_pair(520, 610, seed=6, dtype=dt_ref)
    _, mov = synthetic.make_pair(520, 610, seed=6, dtype=dt_mov)
    params = dict(num_pyr_lvl=1, use_full_res_img=True, use_dog=use_dog, tile_size=200, overlap=30)
    exp, reports = RO.register(ref, mov, **params)
    reg = make_reg(params)
    reg.ref_img, reg.mov_img = ref, mov
    got = reg.register()
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports]
    assert np.array_equal(got, exp)
    w = Warper()
    w.tile_size, w.overlap = 200, 30
    w.image, w.flow = mov, got
    assert np.array_equal(w.warp(), RO.warp(mov, exp, 200, 30))
    fc = TileFlowCalc()
    fc.tile_size, fc.overlap, fc.win_size, fc.num_iter = 200, 30, 29, 2
    fc.ref_img, fc.mov_img = ref, mov
    assert np.array_equal(fc.calc_flow(), RO.tile_flow(ref, mov, 200, 30, 29, 2))
    assert np.array_equal(farneback(mov, ref, 0, 21, 2),
                          O.calc_optical_flow_farneback(mov.astype(np.float32), ref.astype(np.float32), 21, 2))
    reg.ref_img, reg.mov_img = ref, mov
    assert reg.compat_mov_getter and reg.mov_img is ref     # Q4: the reference's mov_img getter returns the REFERENCE image
    reg.compat_mov_getter = False
    assert reg.mov_img is mov


def test_two_streams_reuse_cached_buffers_without_races(ctx):
    """ma_optflow_register runs the flow-independent dog() calls on a companion stream into buffers of the context's
    cache.  Pairs of different sizes and parameters registered back to back, device resident, never synchronised in
    between (so the cache hands buffers of one call to the other stream of the next while kernels are still in flight):
    every repetition must reproduce the first result of its pair bit for bit, and that result is the oracle's."""
    cases = [((700, 900), np.float32, dict(num_pyr_lvl=3, use_full_res_img=True, use_dog=True, tile_size=200, overlap=40)),
             ((1300, 1100), np.uint8, dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=400, overlap=60)),
             ((512, 512), np.uint16, dict(num_pyr_lvl=2, use_full_res_img=False, use_dog=False, tile_size=1000, overlap=100)),
             ((900, 700), np.float32, dict(num_pyr_lvl=1, use_full_res_img=True, use_dog=True, tile_size=300, overlap=50))]
    pairs = [synthetic.make_pair(h, w, 3 + k, dt) for k, ((h, w), dt, _) in enumerate(cases)]
    dev = [(ctx.asdevice(r), ctx.asdevice(m)) for r, m in pairs]
    first = [None] * len(cases)
    pending = []
    for rep in range(6):
        for k in (0, 1, 2, 3, 1, 0, 3, 2)[rep % 2:]:
            reg = make_reg(cases[k][2])
            reg.ref_img, reg.mov_img = dev[k]
            pending.append((k, reg.register()))          # a DeviceArray: nothing waits for the kernels here
    for k, flow in pending:
        got = flow.numpy()
        if first[k] is None:
            first[k] = got
        assert np.array_equal(got, first[k]), k
    for k in (0, 2):
        exp, _ = RO.register(pairs[k][0], pairs[k][1], **cases[k][2])
        assert np.array_equal(first[k], exp)


def test_a_plain_c_host_drives_the_whole_path(tmp_path):
    """tests/c_host/register_host.c: a C program that knows only include/microaligner_hip.h (compiled here with gcc, linked
    against libmicroaligner_hip.so) registers and warps a pair through ma_optflow_register + ma_warp_tiled; its flow,
    warped image and per-level reports equal the oracle's."""
    import subprocess
    from microaligner_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "register_host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_host", "register_host.c"), "-o", str(exe), "-L", libdir,
                    "-lmicroaligner_hip", f"-Wl,-rpath,{libdir}"], check=True, capture_output=True, text=True)
    params = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=180, overlap=36)
    ref, mov = synthetic.make_pair(520, 610, seed=5, dtype=np.uint16)
    ref.tofile(tmp_path / "ref.bin")
    mov.tofile(tmp_path / "mov.bin")
    r = subprocess.run([str(exe), "520", "610", "1", str(tmp_path / "ref.bin"), str(tmp_path / "mov.bin"),
                        str(tmp_path / "flow.bin"), str(tmp_path / "warped.bin"), "2", "1", "1", "180", "36"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    flow = np.fromfile(tmp_path / "flow.bin", np.float32).reshape(520, 610, 2)
    warped = np.fromfile(tmp_path / "warped.bin", np.uint16).reshape(520, 610)
    exp, reports = RO.register(ref, mov, **params)
    assert np.array_equal(flow, exp) and np.array_equal(warped, RO.warp(mov, exp, 180, 36))
    lines = [ln.split() for ln in r.stdout.strip().splitlines()]
    assert [(int(l[0]), int(l[5]) == 1) for l in lines] == [(rep[0], rep[3]) for rep in reports]
    np.testing.assert_allclose([[float(l[3]), float(l[4])] for l in lines], [rep[1:3] for rep in reports], rtol=0, atol=1e-12)


def test_c_register_errors():
    reg = make_reg(dict(num_pyr_lvl=3))
    reg.ref_img = reg.mov_img = np.ones((150, 150), np.float32)      # no level keeps 100 px
    with pytest.raises(ValueError, match="too small"):
        reg.register()
    reg = make_reg(dict(num_pyr_lvl=0, use_full_res_img=False))
    reg.ref_img = reg.mov_img = np.ones((300, 300), np.float32)
    with pytest.raises(ValueError, match="use_full_res_img is False"):
        reg.register()
    reg = make_reg(dict(num_pyr_lvl=-1))
    reg.ref_img = reg.mov_img = np.ones((300, 300), np.float32)
    with pytest.raises(ValueError, match="cannot be less than 0"):
        reg.register()
    # the C entry itself: a reports buffer that is too small, NULL arguments
    import ctypes as C
    from microaligner_amd import _lib as L
    from microaligner_amd.device import get_context
    ctx = get_context()
    d = ctx.asdevice(np.ones((300, 300), np.float32))
    flow = ctx.empty((300, 300, 2), np.float32)
    prm = L.MaParams()
    ctx.lib.ma_params_default(C.byref(prm))
    reps, n = (L.MaLevelReport * 1)(), C.c_int(0)
    assert ctx.lib.ma_optflow_register(ctx.handle, d.ptr, d.ptr, L.MA_F32, 300, 300, C.byref(prm), flow.ptr, reps, 0,
                                       C.byref(n)) == L.MA_EINVAL and b"reports buffer" in ctx.lib.ma_last_error()
    assert ctx.lib.ma_optflow_register(ctx.handle, d.ptr, None, L.MA_F32, 300, 300, C.byref(prm), flow.ptr, None, 0,
                                       None) == L.MA_EINVAL
    # reports are optional
    assert ctx.lib.ma_optflow_register(ctx.handle, d.ptr, d.ptr, L.MA_F32, 300, 300, C.byref(prm), flow.ptr, None, 0,
                                       None) == L.MA_OK
    reg = make_reg(dict(engine="fortran"))
    reg.ref_img = reg.mov_img = np.ones((300, 300), np.float32)
    with pytest.raises(ValueError, match="unknown engine"):
        reg.register()


def test_device_resident_inputs_stay_on_device(ctx):
    ref, mov = synthetic.make_pair(420, 404, 1)
    reg = make_reg(dict(num_pyr_lvl=2, use_full_res_img=True, tile_size=100, overlap=20))
    reg.ref_img, reg.mov_img = ctx.asdevice(ref), ctx.asdevice(mov)
    flow = reg.register()
    assert isinstance(flow, DeviceArray) and flow.shape == (420, 404, 2)
    reg.ref_img, reg.mov_img = ref, mov
    assert np.array_equal(flow.numpy(), reg.register())
    w = Warper()
    w.tile_size, w.overlap = 100, 20
    w.image, w.flow = ctx.asdevice(mov), flow
    assert isinstance(w.warp(), DeviceArray)


def test_recovers_the_synthetic_displacement():
    H, W = 1024, 1024
    ref, mov = synthetic.make_pair(H, W, 5)
    reg = make_reg(dict(num_pyr_lvl=2, use_full_res_img=True, tile_size=400, overlap=60))
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    assert all(r.accepted for r in reg.level_reports)
    dx, dy = synthetic.displacement(H, W)
    true = np.stack([dx, dy], -1)[150:-150, 150:-150]
    err = np.abs(flow[150:-150, 150:-150] - true)
    assert err.mean() < 0.25 and np.percentile(err, 99) < 1.0
    w = Warper()
    w.tile_size, w.overlap = 400, 60
    w.image, w.flow = mov, flow
    warped = w.warp()
    before = np.abs(mov[150:-150, 150:-150] - ref[150:-150, 150:-150]).mean()
    after = np.abs(warped[150:-150, 150:-150] - ref[150:-150, 150:-150]).mean()
    assert after < 0.5 * before


def test_full_size_properties_4096(ctx):
    """BASELINE cfg2 size (4096^2, 3 levels, 25 windows of 1200^2 at full resolution): properties that do not
    need the oracle at this size."""
    H = W = 4096
    ref, _ = synthetic.make_pair(H, W, 2)
    dref = ctx.asdevice(ref)
    # identical images: every window's flow is exactly zero except the L-shaped band of width iters*m+1
    flow = ctx.farneback(dref, dref, 99, 3, tile=1000, overlap=100).numpy()
    assert flow.shape == (H, W, 2)
    # the band (magnitude <~0.3 px at the very edge) leaks ~1e-7 px into the kept centres at overlap 100
    assert np.abs(flow[:3000, :3000]).max() < 1e-5
    assert np.all(flow[:800, :800] == 0)
    assert np.abs(flow).max() < 1.0
    # warp with a zero flow is an exact copy; with an integer flow an exact shifted copy inside each window
    z = ctx.zeros((H, W, 2), np.float32)
    assert np.array_equal(ctx.warp(dref, z, 1000, 100).numpy(), ref)
    shift = np.zeros((H, W, 2), np.float32)
    shift[..., 0] = 7
    shift[..., 1] = -3
    out = ctx.warp(dref, ctx.asdevice(shift), 1000, 100).numpy()
    assert np.array_equal(out[:-3, 7:], ref[3:, :-7])
    # pyrDown of a constant stays constant; NMI(x, x) == 1 on every chunk
    c = ctx.asdevice(np.full((H, W), 200, np.uint8))
    assert np.all(ctx.pyr_down(c).numpy() == 200)
    d = ctx.dog_u8(dref)
    s = ctx.nmi_scores(d, d, 1000 * 1000)
    assert s.shape == (17,) and np.allclose(s, 1.0, atol=1e-12)


def _window(arr, ty, tx, tile, overlap):
    """The zero-padded window (ty, tx) of `arr` as slicer.py cuts it."""
    H, W = arr.shape[:2]
    P = tile + 2 * overlap
    y0, x0 = ty * tile - overlap, tx * tile - overlap
    win = np.zeros((P, P) + arr.shape[2:], arr.dtype)
    ys, xs, ye, xe = max(y0, 0), max(x0, 0), min(y0 + P, H), min(x0 + P, W)
    win[ys - y0:ye - y0, xs - x0:xe - x0] = arr[ys:ye, xs:xe]
    return win


def test_full_size_16384_windows_match_the_oracle_bit_for_bit(ctx):
    """BASELINE cfg3 size (16384^2 float32, 17 x 17 windows of 1200^2).  Windows are independent, so the stitched
    full-size result restricted to one tile must equal the oracle run on that window alone: checked bit for bit on
    interior, edge and corner windows for the Farneback flow and for the warp, plus determinism of a second run."""
    H = W = 16384
    tile, ov = 1000, 100
    ref, mov = synthetic.make_pair(H, W, 3)
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    dflow = ctx.farneback(dmov, dref, 99, 3, tile=tile, overlap=ov)
    flow = dflow.numpy()
    picks = [(0, 0), (8, 5), (3, 16), (16, 9), (16, 16)]   # corner, interior, right edge, bottom edge, last (384 px valid)
    prev = np.stack([_window(mov, ty, tx, tile, ov) for ty, tx in picks])
    nxt = np.stack([_window(ref, ty, tx, tile, ov) for ty, tx in picks])
    exp = RO.O.farneback_batch(prev, nxt, 99, 3, nthreads=len(picks))
    for (ty, tx), e in zip(picks, exp):
        vh, vw = min(tile, H - ty * tile), min(tile, W - tx * tile)
        got = flow[ty * tile:ty * tile + vh, tx * tile:tx * tile + vw]
        assert np.array_equal(got, e[ov:ov + vh, ov:ov + vw]), (ty, tx)
    # warp with the computed flow: same windows, oracle remap of the window
    warped = ctx.warp(dmov, dflow, tile, ov).numpy()
    for ty, tx in picks:
        im, fl = _window(mov, ty, tx, tile, ov), _window(flow, ty, tx, tile, ov)
        m = np.negative(fl)
        m[:, :, 0] += np.arange(m.shape[1])
        m[:, :, 1] += np.arange(m.shape[0]).reshape(-1, 1)
        e = RO.O.remap(im, m)
        vh, vw = min(tile, H - ty * tile), min(tile, W - tx * tile)
        assert np.array_equal(warped[ty * tile:ty * tile + vh, tx * tile:tx * tile + vw], e[ov:ov + vh, ov:ov + vw]), (ty, tx)
    # determinism
    assert sha(ctx.farneback(dmov, dref, 99, 3, tile=tile, overlap=ov).numpy()) == sha(flow)
    # the flow follows the synthetic displacement away from the image border
    ys, xs = np.arange(6000, 8000, dtype=np.float64)[:, None], np.arange(9000, 11000, dtype=np.float64)[None, :]
    dx = synthetic.GLOBAL_SHIFT[0] + 2.0 * np.sin(2 * np.pi * ys / H * 3) + 0 * xs   # synthetic.displacement on the block
    dy = synthetic.GLOBAL_SHIFT[1] + 2.0 * np.cos(2 * np.pi * xs / W * 2) + 0 * ys
    err = np.abs(flow[6000:8000, 9000:11000] - np.stack([dx, dy], -1))
    assert err.mean() < 0.3


def test_image_too_small_for_any_level_is_a_value_error():
    ref, mov = synthetic.make_pair(150, 150, 1)
    reg = make_reg(dict(num_pyr_lvl=3, use_full_res_img=False))
    reg.ref_img, reg.mov_img = ref, mov
    with pytest.raises(ValueError, match="too small"):
        reg.register()


def test_all_zero_moving_image_and_noncontiguous_inputs(ctx):
    ref, _ = synthetic.make_pair(300, 280, 3)
    zero = np.zeros_like(ref)
    params = dict(num_pyr_lvl=1, use_full_res_img=True, use_dog=True, tile_size=100, overlap=14)
    exp, reports = RO.register(ref, zero, **params)
    reg = make_reg(params)
    reg.ref_img, reg.mov_img = ref, zero
    got = reg.register()
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports]
    assert np.array_equal(got, exp)
    # non-contiguous views are accepted like any ndarray
    big_r, big_m = synthetic.make_pair(300, 560, 4)
    reg.ref_img, reg.mov_img = big_r[:, ::2], big_m[:, ::2]
    exp2, _ = RO.register(np.ascontiguousarray(big_r[:, ::2]), np.ascontiguousarray(big_m[:, ::2]), **params)
    assert np.array_equal(reg.register(), exp2)


def test_page_warp_driver_matches_single_page_warps(ctx):
    """one device-resident flow applied to many host pages (u16, the pipeline's page dtype)."""
    H, W = 333, 290
    rng = np.random.default_rng(5)
    pages = [rng.integers(0, 65535, (H, W)).astype(np.uint16) for _ in range(7)]
    from scipy.ndimage import gaussian_filter
    flow = np.stack([gaussian_filter(rng.standard_normal((H, W)), 6) * 40, gaussian_filter(rng.standard_normal((H, W)), 6) * 40],
                    -1).astype(np.float32)
    w = Warper()
    w.tile_size, w.overlap = 100, 12
    w.flow = flow
    out = w.warp_pages(pages)
    assert len(out) == 7
    for p, o in zip(pages, out):
        assert o.dtype == np.uint16 and np.array_equal(o, RO.warp(p, flow, 100, 12))
    # caller-provided output rows (the memmapped TIFF in the pipeline), second call reuses the resident flow
    dst = np.zeros((7, H, W), np.uint16)
    w.warp_pages(pages, out=[dst[i] for i in range(7)])
    assert np.array_equal(dst, np.stack(out))
    assert w.warp_pages([]) == []
    with pytest.raises(ValueError):
        w.warp_pages([pages[0][:10]])


def test_page_warp_driver_starts_behind_kernels_that_still_use_the_workspace(ctx):
    """The driver's device slots live in the context workspace.  A tiled Farneback that was only ENQUEUED (device arrays in,
    device array out, no synchronisation) keeps its 20 planes per window there; page uploads that follow at once must not
    land in them before its kernels are done (the upload engine is ordered behind the compute stream)."""
    H = W = 1500
    ref, mov = synthetic.make_pair(H, W, 77)
    d_ref, d_mov = ctx.asdevice(ref), ctx.asdevice(mov)
    tile, ov, win = 500, 50, 49
    expected = ctx.farneback(d_mov, d_ref, win, 3, tile=tile, overlap=ov).numpy()      # synchronised by the download
    rng = np.random.default_rng(3)
    pages = [rng.integers(0, 65535, (H, W)).astype(np.uint16) for _ in range(6)]
    flow = ctx.asdevice(np.zeros((H, W, 2), np.float32))
    for _ in range(3):
        d_flow = ctx.farneback(d_mov, d_ref, win, 3, tile=tile, overlap=ov)              # enqueued, still running ...
        out = ctx.warp_pages(pages, flow, tile, ov)                                      # ... when the pages arrive
        assert np.array_equal(d_flow.numpy(), expected)
        for p, o in zip(pages, out):
            assert np.array_equal(p, o)                                                  # zero flow: the identity
    with pytest.raises(ValueError):
        ctx.warp_pages(pages, ctx.asdevice(np.zeros((H, W, 2), np.float64)), tile, ov)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32])
@pytest.mark.parametrize("H,W,tile,ov", [(333, 290, 100, 12), (407, 130, 50, 40), (200, 300, 64, 0), (96, 257, 0, 0),
                                         (301, 99, 300, 30)])
def test_page_warp_driver_in_bands_of_one_tile_row(ctx, dtype, H, W, tile, ov):
    """The driver's unit is a band of whole tile rows (an output pixel reads only its own window, warper.py:29-76): with the
    band size forced down to one tile row, every band boundary, the ragged last band, overlaps larger than half a tile and
    the untiled warp (one band) against the oracle -- displacements large enough to reach across the window borders."""
    from microaligner_amd import _lib as L
    rng = np.random.default_rng(H + tile)
    pages = [(rng.random((H, W)) * (255 if dtype == np.uint8 else 60000)).astype(dtype) for _ in range(4)]
    from scipy.ndimage import gaussian_filter
    flow = np.stack([gaussian_filter(rng.standard_normal((H, W)), 5) * 150, gaussian_filter(rng.standard_normal((H, W)), 5) * 150],
                    -1).astype(np.float32)
    before = ctx.get_option(L.MA_OPT_WARP_BAND_BYTES)
    assert before == 32 << 20
    ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, 1)
    try:
        w = Warper()
        w.tile_size, w.overlap = tile, ov
        w.flow = flow
        ctx.transfer_stats(reset=True)
        out = w.warp_pages(pages)
        up, down = ctx.transfer_stats()
    finally:
        ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, before)
    def expected(p):
        if tile:
            return RO.warp(p, flow, tile, ov)
        mp = np.negative(flow)                      # one window: the whole page
        mp[:, :, 0] += np.arange(W)
        mp[:, :, 1] += np.arange(H).reshape(-1, 1)
        return RO.O.remap(p, mp)
    for p, o in zip(pages, out):
        assert np.array_equal(o, expected(p))
    # every source and result row crosses the link exactly once, however the page is cut
    nb = pages[0].nbytes
    assert up == flow.nbytes + 4 * nb and down == 4 * nb
    # the default band size gives the same pages
    w.flow = flow
    for a, b in zip(out, w.warp_pages(pages)):
        assert np.array_equal(a, b)


def test_page_warp_driver_takes_page_locked_and_pageable_buffers_page_by_page(ctx):
    """Every page and every result may be page-locked (Context.host_empty: copied by the DMA engine directly) or pageable
    (staged through the engines' chunks) independently; sizes above the staging threshold (4 MiB) so that both paths run,
    several staging chunks per page with the chunk boundaries falling inside bands."""
    H, W = 2600, 1500                                   # 7.8 MB uint16 pages
    rng = np.random.default_rng(21)
    flow = np.stack([synthetic.displacement(H, W)[0] * 4, synthetic.displacement(H, W)[1] * 4], -1).astype(np.float32)
    pages, outs = [], []
    for k in range(5):
        p = rng.integers(0, 65535, (H, W)).astype(np.uint16)
        if k % 2:
            q = ctx.host_empty((H, W), np.uint16, limit=8)
            q[...] = p
            p = q
        pages.append(p)
        outs.append(ctx.host_empty((H, W), np.uint16, limit=8) if k in (0, 1, 4) else np.empty((H, W), np.uint16))
    from microaligner_amd import _lib as L
    w = Warper()
    w.tile_size, w.overlap = 400, 60
    w.flow = flow
    ref = [ctx.warp(ctx.asdevice(np.array(p)), ctx.asdevice(flow), 400, 60).numpy() for p in pages]
    for band in (1, 32 << 20):
        ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, band)
        try:
            for o in outs:
                o[...] = 0
            w.warp_pages(pages, out=outs)
        finally:
            ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, 32 << 20)
        for k, (o, r) in enumerate(zip(outs, ref)):
            assert np.array_equal(o, r), (band, k)
    assert np.array_equal(ref[0], RO.warp(np.array(pages[0]), flow, 400, 60))


def test_warp_of_a_large_host_page_takes_the_banded_driver(ctx, monkeypatch):
    """Warper.warp() of a host page that is not in HBM yet (the reference's per-page loop): through the page-warp driver
    (bands of tile rows), same pixels as the resident path; a page that IS resident keeps the plain device warp."""
    H, W = 1100, 1000                           # above the resident cache's minimum size
    rng = np.random.default_rng(8)
    page = rng.integers(0, 65535, (H, W)).astype(np.uint16)
    dx, dy = synthetic.displacement(H, W)
    flow = np.stack([dx, dy], -1).astype(np.float32) * 3
    exp = RO.warp(page, flow, 150, 20)
    from microaligner_amd import _lib as L
    monkeypatch.setattr(Warper, "HOST_BANDED_MIN", 1 << 16)
    calls = []
    orig = type(ctx).warp_pages
    monkeypatch.setattr(type(ctx), "warp_pages", lambda self, *a, **k: (calls.append(len(a[0])), orig(self, *a, **k))[1])
    ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, 1)
    try:
        w = Warper()
        w.tile_size, w.overlap = 150, 20
        w.image, w.flow = page.copy(), flow
        got = w.warp()
        assert calls == [1] and isinstance(got, np.ndarray) and np.array_equal(got, exp)
        assert len(w.image) == 0 and len(w.flow) == 0          # inputs consumed like the reference (warper.py:41,45)
        page.setflags(write=False)                             # an immutable host array is remembered once uploaded
        dpage = ctx.asdevice(page)                             # now resident: no driver, no second upload
        w.image, w.flow = page, flow
        ctx.transfer_stats(reset=True)
        got2 = w.warp()
        assert calls == [1] and np.array_equal(got2, exp)
        assert ctx.transfer_stats()[0] == flow.nbytes          # the (writable) flow travels, the page does not
        del dpage
    finally:
        ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, 32 << 20)


def test_sharded_page_warps_fill_the_callers_rows(ctx, tmp_path):
    """parallel.warp_pages: pages dealt to the ranks (one rank here), loaders evaluated by their owner, results written in
    place into the rows of an array every rank can map (a .npy memory map: the output file of the pipeline)."""
    from microaligner_amd import parallel
    H, W = 410, 530
    rng = np.random.default_rng(11)
    pages = [rng.integers(0, 65535, (H, W)).astype(np.uint16) for _ in range(5)]
    from scipy.ndimage import gaussian_filter
    flow = np.stack([gaussian_filter(rng.standard_normal((H, W)), 5) * 30 for _ in range(2)], -1).astype(np.float32)
    exp = [RO.warp(p, flow, 150, 20) for p in pages]
    got = parallel.warp_pages(pages, flow, 150, 20)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    called = []
    store = np.lib.format.open_memmap(tmp_path / "pages.npy", mode="w+", dtype=np.uint16, shape=(5, H, W))
    res = parallel.warp_pages([(lambda k=k: (called.append(k), pages[k])[1]) for k in range(5)], flow, 150, 20, out=store)
    assert res is store and called == [0, 1, 2, 3, 4]
    del res, store
    assert np.array_equal(np.load(tmp_path / "pages.npy"), np.stack(exp))


def test_reference_shaped_numpy_loop_uploads_nothing_twice(ctx):
    """The reference's own statements, numpy in and numpy out (microaligner/__main__.py:418-433 and the per-page loop
    of warp_and_save_pages :296-301): register(), then Warper.warp() of the moving image, then the same flow set on the
    warper for each of 8 uint16 pages.  Results equal the oracle; the context's transfer counters show that the flow is
    never uploaded (register() produced it, the host copy it returned is recognised) and the moving image only once."""
    params = dict(num_pyr_lvl=2, use_full_res_img=True, tile_size=200, overlap=40)
    H, W = 700, 620
    ref, mov = synthetic.make_pair(H, W, seed=9)
    rng = np.random.default_rng(2)
    pages = [rng.integers(0, 65535, (H, W)).astype(np.uint16) for _ in range(8)]
    ctx.trim()
    ctx.sync()
    ctx.transfer_stats(reset=True)
    reg = make_reg(params)
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    # results are handed out read-only (that is what makes recognising them sound): an in-place edit raises
    assert isinstance(flow, np.ndarray) and not flow.flags.writeable
    with pytest.raises(ValueError, match="read-only"):
        flow[10:20, 10:20] = 0
    w = Warper()
    w.tile_size, w.overlap = 200, 40
    w.image, w.flow = mov, flow                     # __main__.py:421-424
    warped = w.warp()
    out = []
    for page in pages:                              # __main__.py:296-301
        w.image = page
        w.flow = flow
        out.append(w.warp())
    up, down = ctx.transfer_stats()
    page_bytes = sum(p.nbytes for p in pages)
    # the flow is never uploaded; the caller's WRITABLE mov array cannot be trusted to be unchanged and goes up twice
    assert up == ref.nbytes + 2 * mov.nbytes + page_bytes, f"uploaded {up} bytes"
    assert down == flow.nbytes + warped.nbytes + page_bytes
    exp_flow, _ = RO.register(ref, mov, **params)
    assert np.array_equal(flow, exp_flow) and np.array_equal(warped, RO.warp(mov, exp_flow, 200, 40))
    for p, o in zip(pages, out):
        assert o.dtype == np.uint16 and np.array_equal(o, RO.warp(p, exp_flow, 200, 40))
    # a caller that marks its inputs read-only gets them recognised as well: mov goes up once
    ctx.forget_host_arrays()
    ctx.transfer_stats(reset=True)
    ref.flags.writeable = mov.flags.writeable = False
    reg.ref_img, reg.mov_img = ref, mov
    flow2 = reg.register()
    w.image, w.flow = mov, flow2
    warped2 = w.warp()
    assert ctx.transfer_stats()[0] == ref.nbytes + mov.nbytes
    assert np.array_equal(flow2, exp_flow) and np.array_equal(warped2, warped)
    # an edited COPY of the flow is a new array: uploaded, and the result follows the edit (the advisor's scenario:
    # flow[y0:y1, x0:x1] = 0 between register() and warp() can no longer pick up a stale device copy)
    edited = flow.copy()
    edited[100:200, 100:200] = 0
    ctx.transfer_stats(reset=True)
    w.image, w.flow = pages[0], edited
    again = w.warp()
    assert ctx.transfer_stats()[0] == edited.nbytes + pages[0].nbytes
    assert np.array_equal(again, RO.warp(pages[0], edited, 200, 40))
    n_before = len(ctx._resident.entries)
    del flow, flow2, exp_flow, w
    import gc
    gc.collect()
    assert len(ctx._resident.entries) < n_before      # the pair dies with the host array


def test_resident_cache_modes(monkeypatch):
    """MICROALIGNER_RESIDENT: results (default; read-only results are recognised, writable caller arrays are not), off
    (everything uploaded every time, writable results), sampled (opt-in, the unsound CRC guard of round 3)."""
    from microaligner_amd.device import Context, use_context
    img, _ = synthetic.make_pair(600, 600, seed=3)
    flow = np.full((600, 600, 2), 1.5, np.float32)
    for mode, uploads, writable in (("results", 2, False), ("readonly", 2, False), ("off", 2, True), ("sampled", 1, True)):
        monkeypatch.setenv("MICROALIGNER_RESIDENT", mode)
        c = Context(0)
        try:
            with use_context(c):
                w = Warper()
                for _ in range(2):
                    w.image, w.flow = img, flow
                    res = w.warp()
                assert c.transfer_stats()[0] == uploads * (img.nbytes + flow.nbytes), mode
                assert res.flags.writeable == writable
                assert np.array_equal(res, RO.warp(img, flow, 1000, 100))
                # a result handed back in: recognised in "results" (and "sampled"), uploaded in "off"
                c.transfer_stats(reset=True)
                w.image, w.flow = res, flow
                w.warp()
                assert c.transfer_stats()[0] == flow.nbytes * (mode != "sampled") + res.nbytes * (mode == "off"), mode
                # a caller's `out` array is filled but its flags are never touched
                mine = np.empty((600, 600), np.float32)
                c.asdevice(img).numpy(out=mine)
                assert mine.flags.writeable and np.array_equal(mine, img)
        finally:
            c.close()
    monkeypatch.setenv("MICROALIGNER_RESIDENT", "sometimes")
    with pytest.raises(ValueError):
        Context(0)


def test_stream_pairs_equals_the_one_pair_path_and_moves_every_byte_once(ctx):
    """parallel.stream_pairs (three engines of one context: upload of pair k+1 and download of pair k-1 under the kernels
    of pair k) against register() + warp() pair by pair: flows, warped images and per-level reports identical, results in
    input order, every byte crosses the bus exactly once (ma_ctx_transfer_stats), shapes and dtypes may change
    mid-stream, and the results are the oracle's."""
    from microaligner_amd import parallel
    params = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=300, overlap=40)
    specs = [((700, 900), np.float32)] * 4 + [((640, 520), np.uint8)] * 3 + [((700, 900), np.float32)] * 2
    pairs = [synthetic.make_pair(h, w, 40 + k, dt) for k, ((h, w), dt) in enumerate(specs)]
    single = []
    for ref, mov in pairs:
        reg = make_reg(params)
        reg.ref_img, reg.mov_img = ref, mov
        flow = reg.register()
        w = Warper()
        w.tile_size, w.overlap = 300, 40
        w.image, w.flow = mov, flow
        single.append((flow, w.warp(), [(r.factor, r.mi_after, r.mi_before, r.accepted) for r in reg.level_reports]))
    ctx.forget_host_arrays()
    stats = {}
    got = list(parallel.stream_pairs(iter(pairs), params, warp=True, stats=stats))
    assert [r.index for r in got] == list(range(len(pairs)))
    for r, (flow, warped, reports) in zip(got, single):
        assert np.array_equal(r.flow, flow) and np.array_equal(r.warped, warped)
        assert [(x.factor, x.mi_after, x.mi_before, x.accepted) for x in r.reports] == reports
        assert r.flow.flags.writeable                     # stream results are plain arrays: nothing is cached behind them
    assert stats["h2d_bytes"] == sum(a.nbytes + b.nbytes for a, b in pairs)
    assert stats["d2h_bytes"] == sum(f.nbytes + w_.nbytes for f, w_, _ in single)
    assert stats["pairs"] == len(pairs) and min(stats[k] for k in ("h2d_busy_ms", "compute_busy_ms", "d2h_busy_ms")) > 0
    exp_flow, _ = RO.register(*pairs[5], **params)
    assert np.array_equal(got[5].flow, exp_flow) and np.array_equal(got[5].warped, RO.warp(pairs[5][1], exp_flow, 300, 40))
    # flows only, into caller-provided rows (a memmap in the pipeline)
    dst = np.zeros((4, 700, 900, 2), np.float32)
    for r in parallel.stream_pairs(pairs[:4], params, warp=False, out=lambda i: (dst[i], None)):
        assert r.warped is None
    assert all(np.array_equal(dst[i], single[i][0]) for i in range(4))
    # lanes by default: two when two working sets fit the device comfortably, one when they would not
    assert stats["compute_lanes"] == 2
    import microaligner_amd.device as D
    real_info = D.device_info
    D.device_info = lambda dev=0: dict(real_info(dev), mem_total=1 << 28)     # a device with 256 MiB
    try:
        st1 = {}
        small = list(parallel.stream_pairs(iter(pairs[:3]), params, warp=True, stats=st1))
    finally:
        D.device_info = real_info
    assert st1["compute_lanes"] == 1 and all(np.array_equal(r.flow, single[i][0]) for i, r in enumerate(small))
    # two compute lanes (pairs registered two at a time on two contexts): same bits, same order
    two = list(parallel.stream_pairs(iter(pairs), params, warp=True, compute_lanes=2))
    assert [r.index for r in two] == list(range(len(pairs)))
    assert all(np.array_equal(r.flow, f) and np.array_equal(r.warped, w_) for r, (f, w_, _) in zip(two, single))
    # register_pairs routes host pairs through the stream; loaders are evaluated lazily; out= rows are filled in place
    flows = parallel.register_pairs(pairs[:4], params, warp=True)
    assert all(np.array_equal(f, single[i][0]) and np.array_equal(w_, single[i][1]) for i, (f, w_) in enumerate(flows))
    called = []
    loaders = [(lambda k=k: (called.append(k), pairs[k])[1]) for k in range(4)]
    store = (np.zeros((4, 700, 900, 2), np.float32), np.zeros((4, 700, 900), np.float32))
    assert parallel.register_pairs(loaders, params, warp=True, out=store) is store and called == [0, 1, 2, 3]
    assert all(np.array_equal(store[0][i], single[i][0]) and np.array_equal(store[1][i], single[i][1]) for i in range(4))
    # the engine entry points reject what they cannot serve
    with pytest.raises(ValueError):
        ctx.engine_sync(7)
    with pytest.raises(ValueError):
        ctx.engine_upload(ctx.empty((10,), np.float32), np.zeros(11, np.float32))


def test_companion_stream_switch_changes_nothing_but_the_schedule(ctx):
    """MA_OPT_COMPANION_STREAM off (every kernel alone on the chip, for profiles) and on: same flow, same reports,
    through the C engine; the option reads back; unknown options are rejected."""
    from microaligner_amd import _lib as L
    params = dict(num_pyr_lvl=3, use_full_res_img=True, use_dog=True, tile_size=200, overlap=40)
    ref, mov = synthetic.make_pair(900, 1100, seed=8)
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    out = {}
    assert ctx.companion_stream is True
    try:
        for on in (True, False, True):
            ctx.companion_stream = on
            assert ctx.companion_stream is on
            reg = make_reg(params)
            reg.ref_img, reg.mov_img = dref, dmov
            flow = reg.register().numpy()
            out.setdefault(on, []).append((flow, [(r.factor, r.shape, r.mi_after, r.mi_before, r.accepted)
                                                  for r in reg.level_reports]))
    finally:
        ctx.companion_stream = True
    base_flow, base_rep = out[True][0]
    for flow, rep in out[True][1:] + out[False]:
        assert np.array_equal(flow, base_flow) and rep == base_rep
    exp, _ = RO.register(ref, mov, **params)
    assert np.array_equal(base_flow, exp)
    with pytest.raises(ValueError):
        ctx.set_option(99, 1)
    assert ctx.get_option(L.MA_OPT_WORKSPACE_LIMIT) > 0


def test_register_pairs_with_lanes_matches_single_lane():
    """Several pairs in flight on one GPU (one context and host thread per lane) give the same bits."""
    from microaligner_amd import parallel
    pairs = [synthetic.make_pair(700, 900, seed) for seed in (1, 2, 3, 4, 5)]
    params = dict(num_pyr_lvl=2, use_full_res_img=True, tile_size=300, overlap=40, use_dog=True)
    one = parallel.register_pairs(pairs, params, warp=True, lanes=1)
    many = parallel.register_pairs(pairs, params, warp=True, lanes=3)
    for (f1, w1), (f3, w3) in zip(one, many):
        assert np.array_equal(f1, f3) and np.array_equal(w1, w3)


def test_cycle_chain_matches_the_oracle_pipeline():
    """BASELINE cfg4 in miniature: TCZYX cycles, z max-projection + uint8 normalisation of the reference channel,
    the chained registration (each cycle against the previous warped one) and the page warps, against the same
    pipeline composed from the oracle (numpy maximum, oracle normalize / register / warp)."""
    from oracle import oracle as O
    from microaligner_amd import parallel
    rng = np.random.default_rng(5)
    H, W, C_, Z_ = 420, 510, 2, 3
    base, _ = synthetic.make_pair(H + 40, W + 40, 9)
    cycles = []
    for cyc in range(3):
        dy, dx = 3 * cyc, 2 * cyc                   # every cycle drifts a little further
        stack = np.empty((C_, Z_, H, W), np.uint16)
        for c in range(C_):
            for z in range(Z_):
                img = base[20 + dy:20 + dy + H, 20 + dx:20 + dx + W] * (40.0 + 10 * c) * (1.0 - 0.2 * z)
                stack[c, z] = np.clip(img + rng.normal(0, 20, (H, W)), 0, 65535).astype(np.uint16)
        cycles.append(stack)
    params = dict(num_pyr_lvl=2, use_full_res_img=True, tile_size=200, overlap=30, use_dog=True)
    aligned, flows = parallel.register_cycle_chain(cycles, ref_channel_ids=[1, 1, 1], params=params)

    def conditioned(stack):
        return O.normalize_minmax_u8(np.maximum.reduce(list(stack[1])).astype(np.float32))
    ref = conditioned(cycles[0])
    assert np.array_equal(aligned[0], cycles[0]) and flows[0] is None
    for cyc in (1, 2):
        mov = conditioned(cycles[cyc])
        flow, _ = RO.register(ref, mov, **params)
        assert np.array_equal(flows[cyc], flow)
        ref = RO.warp(mov, flow, 200, 30)
        for c in range(C_):
            for z in range(Z_):
                assert np.array_equal(aligned[cyc][c, z], RO.warp(cycles[cyc][c, z], flow, 200, 30))


@pytest.mark.parametrize("seed", range(48))
def test_register_random_configurations_match_the_oracle(seed):
    """Seeded random shapes, dtypes and parameters (ragged sizes, tiny tiles, odd/even overlaps, with and without
    DOG / full-resolution level, 1-4 iterations): flow, level decisions and warp equal the oracle orchestration."""
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(230, 520)), int(rng.integers(230, 520))
    dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
    tile = int(rng.integers(60, 200))
    overlap = int(rng.integers(8, min(40, tile // 2)))
    params = dict(num_pyr_lvl=int(rng.integers(0, 3)), num_iterations=int(rng.integers(1, 5)), tile_size=tile,
                  overlap=overlap, use_full_res_img=bool(rng.integers(0, 2)), use_dog=bool(rng.integers(0, 2)))
    if params["num_pyr_lvl"] == 0 or min(H, W) / 2 < 100:
        params["use_full_res_img"] = True
    if rng.integers(0, 4) == 0:
        ref, mov = synthetic.make_unrelated_pair(H, W, seed, dtype)
    else:
        ref, mov = synthetic.make_pair(H, W, seed, dtype)
    reg = make_reg(params)
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    exp_flow, reports = RO.register(ref, mov, **params)
    assert [r.accepted for r in reg.level_reports] == [r[3] for r in reports], params
    assert np.array_equal(flow, exp_flow), params
    w = Warper()
    w.tile_size, w.overlap = tile, overlap
    w.image, w.flow = mov, flow
    assert np.array_equal(w.warp(), RO.warp(mov, exp_flow, tile, overlap)), params


def test_results_land_by_dma_in_caller_memory_that_was_page_locked_in_place(ctx, tmp_path):
    """ma_host_register (device.host_register, parallel.shared_array): a caller-owned array -- here an np.empty and a
    shared-memory .npy memmap like the ones a multi-rank run writes its results into -- is page-locked in place; downloads
    into it and uploads from it go by DMA directly (the staging rings stay untouched: nothing is counted twice, the bytes are
    right), and the registration ends with the array."""
    from microaligner_amd import device, parallel
    rng = np.random.default_rng(8)
    src = rng.random((2048, 3000)).astype(np.float32)            # 24.6 MB: above the staging threshold
    dev = ctx.asdevice(src)
    plain = np.empty_like(src)
    assert device.host_register(plain) is True
    dev.numpy(out=plain)
    assert np.array_equal(plain, src)
    back = ctx.asdevice(plain.copy()).numpy()
    assert np.array_equal(back, src)
    # a node-wide result array: one rank here, the mapping is page-locked where a device exists
    name = f"ma_test_{os.getpid()}.npy"
    shared = parallel.shared_array(name, (2,) + src.shape, np.float32, directory="/dev/shm", unlink=True)
    assert parallel.arr_is_page_locked(shared)
    dev.numpy(out=shared[1])
    assert np.array_equal(shared[1], src) and not shared[0].any()
    # upload FROM page-locked caller memory
    shared[0][...] = src[::-1]
    assert np.array_equal(ctx.asdevice(shared[0]).numpy(), src[::-1])
    del shared, plain
    import gc
    gc.collect()
    # memory that cannot be registered is reported, not raised
    assert device.host_register(np.empty(0, np.float32)) is False


def test_a_copy_that_runs_past_a_page_locked_range_is_staged(ctx):
    """is_page_locked (csrc/ma_api.hip) decides per TRANSFER, not per start pointer: a range that begins inside a registered row
    (or inside a hipHostMalloc buffer) and ends beyond it is partly pageable and must take the staging ring; the bytes arrive
    either way.  parallel.shared_array registers row by row in multi-rank runs, so `arr[r:r + 2]` with only row r registered
    is exactly this case."""
    import ctypes as C
    from microaligner_amd import device
    from microaligner_amd import _lib as L
    lib = L.load()

    def direct(ptr, nbytes):
        out = C.c_int(-1)
        L.check(lib.ma_host_transfer_is_direct(C.c_void_p(ptr), C.c_size_t(nbytes), C.byref(out)))
        return out.value

    rng = np.random.default_rng(9)
    rows = np.empty((3, 1024, 2048), np.float32)                 # 8 MB per row: above the staging threshold
    row_bytes = rows[0].nbytes
    base = rows.ctypes.data
    assert direct(base, row_bytes) == 0                          # pageable
    assert lib.ma_host_register(C.c_void_p(base + row_bytes), C.c_size_t(row_bytes)) == L.MA_OK      # row 1 only
    try:
        assert direct(base + row_bytes, row_bytes) == 1
        assert direct(base + row_bytes + 4096, row_bytes - 4096) == 1
        assert direct(base + row_bytes, row_bytes + 1) == 0      # starts inside, ends one byte past the registration
        assert direct(base + row_bytes, 2 * row_bytes) == 0
        assert direct(base, 2 * row_bytes) == 0                  # starts before it
        src = rng.random((2, 1024, 2048)).astype(np.float32)
        dev = ctx.asdevice(src)
        dev.numpy(out=rows[1:3])                                 # download across the end of the registered row
        assert np.array_equal(rows[1:3], src)
        assert np.array_equal(ctx.asdevice(rows[1:3].copy()).numpy(), src)
        up = ctx.empty(src.shape, np.float32)                    # upload FROM the overrunning range
        L.check(lib.ma_memcpy_h2d(ctx.handle, up.ptr, C.c_void_p(base + row_bytes), C.c_size_t(2 * row_bytes)))
        assert np.array_equal(up.numpy(), src)
    finally:
        assert lib.ma_host_unregister(C.c_void_p(base + row_bytes)) == L.MA_OK
    assert direct(base + row_bytes, row_bytes) == 0
    # the runtime's own page-locked allocations: whole buffer direct, a range past its end not
    pinned = ctx.host_empty((1024, 2048), np.float32)
    assert direct(pinned.ctypes.data, pinned.nbytes) == 1
    assert direct(pinned.ctypes.data, 64 << 20) == 0
    del pinned


def test_two_ranks_download_into_their_own_page_locked_rows(tmp_path):
    """parallel.shared_array with more than one rank: every rank page-locks the rows shard() deals to it, in its own mapping,
    one registration per row; downloads into own rows go by DMA as they are, a range that runs into a foreign row is staged,
    everybody sees everybody's rows.  Two processes over gloo sharing the box's device (tests/_dist_gpu_worker.py)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "res.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "_dist_gpu_worker.py"), str(out)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert [row["rank"] for row in res] == [0, 1]
    for row in res:
        assert row["ok"] and row["locked"]
        assert all(row["direct_own"]) and not any(row["direct_other"]) and row["direct_span"] is False


def test_transient_page_locking_of_big_pageable_arrays(ctx):
    """MICROALIGNER_TRANSIENT_PIN=1 (opt-in): a blocking copy of a pageable array of
    >= 64 MiB page-locks the caller's range for its own duration -- hipHostRegister, one DMA, hipHostUnregister -- instead of
    staging it.  In a child process (the policy is read once per process): bytes arrive in both directions, the library
    reports the mode, nothing stays registered (a permanent registration of the same array succeeds afterwards), smaller arrays
    and ranges that cannot be registered (part of them registered already) are staged as before."""
    import subprocess
    import sys
    code = r'''
import numpy as np, ctypes as C
from microaligner_amd import device, _lib as L
from microaligner_amd.device import get_context
ctx = get_context()
rng = np.random.default_rng(3)
big = rng.integers(0, 255, (9000, 9000), dtype=np.uint8)                 # 81 MB, pageable
small = rng.integers(0, 255, (3000, 3000), dtype=np.uint8)               # 9 MB: below the threshold
assert device.transfer_mode(big) == "transient" and device.transfer_mode(small) == "staged"
d = ctx.asdevice(big)
back = np.empty_like(big)
d.numpy(out=back)
assert np.array_equal(back, big)
assert device.transfer_mode(big) == "transient"                          # nothing was left registered ...
assert device.host_register(big) is True and device.transfer_mode(big) == "direct"   # ... a permanent registration still works
assert device.host_unregister(big)
# a range that is partly registered already cannot be registered again: staged, bytes still right
half = big[:4500]
assert device.host_register(half) is True
d2 = ctx.empty(big.shape, big.dtype)
L.check(L.load().ma_memcpy_h2d(ctx.handle, d2.ptr, C.c_void_p(big.ctypes.data), C.c_size_t(big.nbytes)))
assert np.array_equal(d2.numpy(), big)
device.host_unregister(half)
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MICROALIGNER_TRANSIENT_PIN="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]
