"""Parity with the REAL OpenCV through reference-held vectors: tests/golden/cv2_4.5.5.npz.

The file is made by tests/golden/make_cv2_golden.py -- numpy + cv2 only, no GPU, no build of this repository -- on ANY
machine where the reference's pin (opencv-contrib-python==4.5.5.64, environment.yaml:75) installs, and committed.  It holds
inputs and outputs of every cv2 call site of the hot path at the reference's own signatures.  Consumers:

  * unmarked tests : the CPU oracle (oracle/ma_oracle.c) against the file          -- any host
  * -m gpu tests   : the HIP path through the C-ABI against the file              -- the GPU box

Bars: integer outputs bit for bit (north_star: "warped integer output is bit-exact to cv2.remap"); flows within FLOW_TOL
px of cv2.calcOpticalFlowFarneback, in whichever of the two muladd models the recorded build follows (reported); float32
images within a few ulp; the dog() chain exact in one of its four rounding models (reported).  While the file is absent
(this image has no cv2: no wheel, no network) they SKIP with the one command that lifts the skip -- and the plumbing of
generator and consumers is exercised all the same, against a throw-away cv2 stand-in that forwards to the oracle
(tests/golden/_cv2_standin): that run pins nothing, it only proves that the day a real file arrives the tests read it right.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
sys.path.insert(0, GOLDEN)
import make_cv2_golden as G          # noqa: E402  (its cv2 import is inside main(): the input helpers need numpy only)

FLOW_TOL = 1e-3      # px
F32_RTOL = 2e-6      # float32 images: a few ulp (association order of vectorised sums)
KINDS = ["farneback", "farneback_window", "remap", "pyr_down", "pyr_up", "dog", "normalize_u8", "warp_affine"]
SKIP_REASON = ("tests/golden/cv2_4.5.5.npz is absent: run `python tests/golden/make_cv2_golden.py` on any machine with "
               "numpy and opencv-contrib-python==4.5.5.64 and commit the file (parity with the real OpenCV stays unpinned "
               "until then)")


def golden_path():
    env = os.environ.get("MA_CV2_GOLDEN")
    if env:
        return env
    files = sorted(glob.glob(os.path.join(GOLDEN, "cv2_*.npz")))
    pinned = [f for f in files if os.path.basename(f) == f"cv2_{G.PINNED}.npz"]
    return (pinned or files or [None])[-1]


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["__meta__"]).decode())
    assert meta["format"] == G.FORMAT
    return z, meta


class Report:
    def __init__(self, who):
        self.who, self.lines = who, []

    def __call__(self, name, got, exp, note=""):
        d = np.abs(got.astype(np.float64) - exp.astype(np.float64))
        self.lines.append(f"[cv2 golden] {self.who} {name}: exact={np.array_equal(got, exp)} max|d|={d.max():.3g} "
                          f"differing={int((d > 0).sum())}/{d.size} {note}")
        print(self.lines[-1])


# ---- the two implementations under test, behind one face ---------------------------------------------------------------
class OracleImpl:
    name = "oracle"

    def __init__(self):
        from oracle import oracle as O
        self.O = O
        self.dog_models = {0: "SSE2 baseline (mul, add)", O.DOG_FUSED_BLUR: "fused GaussianBlur",
                           O.DOG_FUSED_SCALE: "fused normalize", O.DOG_FUSED: "AVX2 + FMA3 objects (both fused)"}

    def farneback(self, mov, ref, win, iters, fused):
        return self.O.calc_optical_flow_farneback(mov, ref, win, iters, fused=fused)

    def remap(self, src, m):
        return self.O.remap(src, m)

    def pyr_down(self, img):
        return self.O.pyr_down(img)

    def pyr_up(self, flow, scale, dst_hw):
        return self.O.pyr_up(flow * scale, dstsize=tuple(dst_hw[::-1]))

    def dog(self, img, flags):
        return self.O.dog(img, flags=flags)

    def normalize_u8(self, img):
        return self.O.normalize_minmax_u8(img)

    def warp_affine(self, img, M, dsize):
        return self.O.warp_affine(img, M, dsize=tuple(dsize))

    def dog_steps(self, img, fused):
        """(normalised image, blur sigma 5, blur sigma 9) of the dog() chain in one rounding model."""
        f = self.O.normalize_minmax_f32(img, 0.0, 1.0, fused=fused)
        return f, self.O.gaussian_blur(f, 41, 5, fused=fused), self.O.gaussian_blur(f, 41, 9, fused=fused)


class HipImpl:
    name = "HIP"

    def __init__(self, ctx):
        from microaligner_amd import _lib as L
        self.ctx = ctx
        self.dog_models = {0: "SSE2 baseline (mul, add)", L.MA_DOG_FUSED_BLUR: "fused GaussianBlur",
                           L.MA_DOG_FUSED_SCALE: "fused normalize",
                           L.MA_DOG_FUSED_BLUR | L.MA_DOG_FUSED_SCALE: "AVX2 + FMA3 objects (both fused)"}

    def farneback(self, mov, ref, win, iters, fused):
        c = self.ctx
        return c.farneback(c.asdevice(mov), c.asdevice(ref), win, iters, fused=fused).numpy()

    def remap(self, src, m):
        return self.ctx.remap(self.ctx.asdevice(src), self.ctx.asdevice(m)).numpy()

    def pyr_down(self, img):
        return self.ctx.pyr_down(self.ctx.asdevice(img)).numpy()

    def pyr_up(self, flow, scale, dst_hw):
        return self.ctx.pyr_up_flow(self.ctx.asdevice(flow), tuple(dst_hw), float(scale)).numpy()

    def dog(self, img, flags):
        return self.ctx.dog_u8(self.ctx.asdevice(img), flags=flags).numpy()

    def normalize_u8(self, img):
        return self.ctx.normalize_minmax_u8(self.ctx.asdevice(img)).numpy()

    def warp_affine(self, img, M, dsize):
        return self.ctx.warp_affine_cv(self.ctx.asdevice(img), M, dsize=tuple(dsize)).numpy()

    dog_steps = None     # the fused kernel keeps the intermediates in LDS


# ---- one check per kind of case ------------------------------------------------------------------------------------------
def _close_or_equal(got, exp, rtol=F32_RTOL, atol=1e-4):
    if np.issubdtype(exp.dtype, np.integer):
        assert got.dtype == exp.dtype and np.array_equal(got, exp)
    else:
        np.testing.assert_allclose(got, exp, rtol=rtol, atol=atol)


def check_kind(impl, kind, z, meta, strict=False):
    """Every case of `kind` in the file through `impl`.  strict: the file was made by the oracle stand-in, so the default
    rounding models must reproduce it bit for bit (plumbing check)."""
    rep = Report(impl.name)
    cases = {n: c for n, c in meta["cases"].items() if c["kind"] == kind}
    assert cases, f"no {kind} case in the file"
    for name, c in sorted(cases.items()):
        if kind in ("farneback", "farneback_window"):
            if kind == "farneback":
                ref, mov, exp = z[c["ref"]], z[c["mov"]], z[name]
            else:
                ref, mov = G.upsample4(z[c["ref"]]), G.upsample4(z[c["mov"]])
                if c["derive"] == "f32":
                    ref, mov = G.frac_f32(ref), G.frac_f32(mov)
                exp = z[c["sample"]]
            got = {f: impl.farneback(mov, ref, c["winsize"], c["iterations"], f) for f in (False, True)}
            if kind == "farneback_window":
                exact = [("fma" if f else "mul+add") for f in got
                         if hashlib.sha256(np.ascontiguousarray(got[f]).tobytes()).hexdigest() == c["sha256"]]
                print(f"[cv2 golden] {impl.name} {name}: whole 1200^2 flow bit-identical in model(s) {exact or 'none'}")
                assert list(got[False].shape) == c["shape"]
                got = {f: g[::c["stride"], ::c["stride"]] for f, g in got.items()}
            for f in got:
                rep(name, got[f], exp, "(fma model)" if f else "(mul+add model)")
            err = {f: float(np.abs(got[f] - exp).max()) for f in got}
            assert min(err.values()) <= FLOW_TOL, err
            if strict:
                assert np.array_equal(got[False], exp)
        elif kind == "remap":
            got = impl.remap(z[c["src"]], z[c["map"]])
            rep(name, got, z[name])
            _close_or_equal(got, z[name], atol=1e-3)
        elif kind == "pyr_down":
            got = impl.pyr_down(z[c["src"]])
            rep(name, got, z[name])
            _close_or_equal(got, z[name])
        elif kind == "pyr_up":
            got = impl.pyr_up(z[c["src"]], c["scale"], c["dst_hw"])
            rep(name, got, z[name])
            _close_or_equal(got, z[name], atol=1e-5)
        elif kind == "dog":
            src, exp = z[c["src"]], z[name]
            exact = []
            for flags, label in impl.dog_models.items():
                got = impl.dog(src, flags)
                rep(name, got, exp, f"[{label}]")
                if np.array_equal(got, exp):
                    exact.append(label)
            print(f"[cv2 golden] {impl.name} {name}: rounding models that reproduce the recorded build bit for bit: {exact}")
            if "norm" in c and impl.dog_steps:
                for fused in (False, True):
                    for step, g in zip(("norm", "blur_lo", "blur_hi"), impl.dog_steps(src, fused)):
                        rep(f"{name} step {step}", g, z[c[step]], "(fma)" if fused else "(mul+add)")
            assert exact, "no rounding model of the dog() chain reproduces the recorded OpenCV build bit for bit"
            if strict:
                assert impl.dog_models[0] in exact
        elif kind == "normalize_u8":
            got = impl.normalize_u8(z[c["src"]])
            rep(name, got, z[name])
            assert np.array_equal(got, z[name])
        elif kind == "warp_affine":
            got = impl.warp_affine(z[c["src"]], z[c["M"]], c["dsize"])
            rep(name, got, z[name])
            _close_or_equal(got, z[name], atol=1e-3)
        else:
            raise AssertionError(kind)
    return rep.lines


# ---- against the committed file --------------------------------------------------------------------------------------------
def _real_file():
    path = golden_path()
    if not path or not os.path.exists(path):
        pytest.skip(SKIP_REASON)
    z, meta = load(path)
    if meta.get("standin"):
        pytest.skip(f"{path} was made with the oracle stand-in, not with OpenCV: it pins nothing. " + SKIP_REASON)
    return z, meta


def test_golden_file_records_the_opencv_build():
    z, meta = _real_file()
    facts = meta["facts"]
    print("\n[cv2 golden] recorded with cv2", facts["cv2_version"], "numpy", facts["numpy_version"],
          "| CPU features in use:", facts["cpu_features_in_use"], "| IPP:", facts["ipp"])
    for ln in facts["build_lines"]:
        print("[cv2 golden]   ", ln)
    assert facts["cv2_version"].split(".")[0] == "4", "the reference pins opencv-contrib-python >=4.5,<5.0 (setup.py:40)"
    if not facts["cv2_version"].startswith(G.PINNED):
        print(f"[cv2 golden] NOTE: not the pinned {G.PINNED} (environment.yaml:75)")


@pytest.mark.parametrize("kind", KINDS)
def test_oracle_matches_the_recorded_opencv(kind):
    z, meta = _real_file()
    check_kind(OracleImpl(), kind, z, meta)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_hip_path_matches_the_recorded_opencv(ctx, kind):
    z, meta = _real_file()
    check_kind(HipImpl(ctx), kind, z, meta)


# ---- plumbing, with the stand-in (runs everywhere, pins nothing) -----------------------------------------------------------
@pytest.fixture(scope="module")
def standin_file(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cv2_standin") / "cv2_standin.npz")
    env = dict(os.environ, PYTHONPATH=os.path.join(GOLDEN, "_cv2_standin") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_cv2_golden.py"), "--out", out], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return out


def test_generator_and_consumers_work_end_to_end_with_a_standin(standin_file):
    """The generator runs as a stand-alone script (numpy + "cv2" only), writes every case, and the consumers read them
    back: with the stand-in the file IS the oracle, so the oracle must reproduce it bit for bit in its default models."""
    z, meta = load(standin_file)
    assert meta["standin"] is True and meta["facts"]["cv2_version"] == "4.5.5"
    assert {c["kind"] for c in meta["cases"].values()} == set(KINDS)
    assert os.path.getsize(standin_file) < 12e6, "keep the fixture small enough to commit"
    for kind in KINDS:
        check_kind(OracleImpl(), kind, z, meta, strict=True)
    # a stand-in file is never mistaken for a pin
    os.environ["MA_CV2_GOLDEN"] = standin_file
    try:
        with pytest.raises(pytest.skip.Exception):
            _real_file()
    finally:
        del os.environ["MA_CV2_GOLDEN"]
    # the generator refuses other OpenCV versions unless told otherwise (checked without importing anything heavy)
    src = open(os.path.join(GOLDEN, "make_cv2_golden.py")).read()
    assert "import microaligner_amd" not in src and "from microaligner_amd" not in src and "from oracle" not in src


@pytest.mark.gpu
def test_hip_consumers_work_end_to_end_with_a_standin(ctx, standin_file):
    z, meta = load(standin_file)
    for kind in KINDS:
        check_kind(HipImpl(ctx), kind, z, meta, strict=True)
