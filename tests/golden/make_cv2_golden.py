#!/usr/bin/env python3
"""Golden vectors from the REAL OpenCV, for every call site of the hot path -- a five-minute job on any machine.

    pip install numpy opencv-contrib-python==4.5.5.64        # the reference's pin, environment.yaml:75
    python tests/golden/make_cv2_golden.py                   # writes tests/golden/cv2_4.5.5.npz
    git add tests/golden/cv2_4.5.5.npz                       # commit it: tests/test_cv2_golden.py stops skipping

Needs nothing but numpy and cv2: it does not import this repository, needs no GPU and no built library, so it runs
on a laptop.  What it records is what the reference's own statements produce (file:line of each call site below):
inputs AND outputs go into one .npz, so the consumers regenerate nothing.

  calcOpticalFlowFarneback   flow_calc.py:33-44      uint8 + float32, winsize 19 / 99, 1 and 3 iterations; one 1200^2 window
  remap                      warper.py:65, optflow_registrator.py:45      uint8 / uint16 / float32 / float32 x 2 channels
  pyrDown                    optflow_registrator.py:194                   three dtypes, even and odd sizes
  pyrUp(dstsize)             optflow_registrator.py:140,150,164,169,212   2-channel float32, even and odd targets
  dog() chain                optflow_registrator.py:249-274               normalize -> GaussianBlur x 2 -> diff -> normalize
  normalize -> uint8         shared_modules/utils.py:94
  warpAffine                 feature_reg/feature_registrator.py:128-132

tests/test_cv2_golden.py compares the CPU oracle (any host) and the HIP path (-m gpu) with the file: integer outputs
bit for bit, flows within 1e-3 px, and reports which rounding model of the window blur / of the dog() chain the
recorded build follows.  The file also keeps cv2.getBuildInformation()'s version / dispatch / IPP lines: they decide
which of the known divergence suspects apply (AVX2 + FMA3 dispatch of filter.simd.hpp, IPP's GaussianBlur).
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PINNED = "4.5.5"
FORMAT = 1


# ---- inputs (stored in the file; nothing here has to be reproducible elsewhere) -----------------------------------------
def texture(cv2, h, w, seed):
    """Smooth, feature-rich float32 image in [0, 255]: blurred white noise, two scales."""
    rs = np.random.RandomState(seed)
    a = cv2.GaussianBlur(rs.standard_normal((h + 40, w + 40)).astype(np.float32), (0, 0), 4.0)
    b = cv2.GaussianBlur(rs.standard_normal((h + 40, w + 40)).astype(np.float32), (0, 0), 1.5)
    t = a + 0.35 * b
    t = (t - t.min()) / (t.max() - t.min()) * 255.0
    return np.ascontiguousarray(t.astype(np.float32))


def make_pair(cv2, h, w, seed):
    """(ref, mov) float32: mov is the canvas resampled at p + d(p), d = shift + smooth field (SURVEY 8d's recipe)."""
    canvas = texture(cv2, h, w, seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    dx = 3.3 + 2.0 * np.sin(2 * np.pi * ys / h * 3)
    dy = -2.1 + 2.0 * np.cos(2 * np.pi * xs / w * 2)
    ref = np.ascontiguousarray(canvas[20:20 + h, 20:20 + w])
    mov = cv2.remap(canvas, (xs + 20 + dx).astype(np.float32), (ys + 20 + dy).astype(np.float32), cv2.INTER_CUBIC)
    return ref, np.ascontiguousarray(np.clip(mov, 0, 255).astype(np.float32))


def to_dtype(img, dtype):
    if dtype == np.float32:
        return img.astype(np.float32)
    if dtype == np.uint16:
        return np.clip(np.rint(img * 257.0), 0, 65535).astype(np.uint16)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def frac_f32(u8):
    """float32 image derived EXACTLY from a uint8 one (so the big float case stores one byte per pixel): value +
    a position-dependent multiple of 1/16."""
    h, w = u8.shape
    ys, xs = np.mgrid[0:h, 0:w]
    return (u8.astype(np.float32) + ((xs * 7 + ys * 13) % 16).astype(np.float32) / 16.0).astype(np.float32)


def upsample4(u8):
    """(h, w) uint8 -> (4h, 4w) uint8 by separable linear interpolation in exact integer arithmetic (edge replicated):
    the 1200^2 Farneback window is stored as its 300^2 seed image and rebuilt bit for bit by generator and consumers."""
    def along0(a):
        a = a.astype(np.int32)
        nxt = np.concatenate([a[1:], a[-1:]], 0)
        return np.stack([(a * (4 - k) + nxt * k + 2) // 4 for k in range(4)], 1).reshape((-1,) + a.shape[1:])
    return np.ascontiguousarray(along0(along0(u8).T).T.astype(np.uint8))


def random_map(h, w, sh, sw, seed):
    rs = np.random.RandomState(seed)
    m = np.empty((h, w, 2), np.float32)
    m[..., 0] = rs.uniform(-3, sw + 3, (h, w))
    m[..., 1] = rs.uniform(-3, sh + 3, (h, w))
    m[::7, ::5, 0] = np.round(m[::7, ::5, 0])            # exact pixel centres
    m[3::11, 2::9, 1] += 1.0 / 64                        # half a quantisation step of the 1/32 px grid
    m[5::13, 1::6] = np.round(m[5::13, 1::6] * 32) / 32 + 1.0 / 64   # exactly on the rounding boundary
    return m


# ---- the reference's statements ------------------------------------------------------------------------------------------
def cv_farneback(cv2, mov, ref, win, iters):
    """flow_calc.py:33-44"""
    return cv2.calcOpticalFlowFarneback(mov, ref, None, pyr_scale=0.5, levels=0, winsize=win, iterations=iters,
                                        poly_n=1, poly_sigma=1.7, flags=cv2.OPTFLOW_FARNEBACK_GAUSSIAN)


def cv_dog(cv2, img, low_sigma=5, high_sigma=9):
    """optflow_registrator.py:249-274; also returns the intermediates the chain goes through"""
    fimg = cv2.normalize(img, None, 0, 1, cv2.NORM_MINMAX, cv2.CV_32F)
    ks = (low_sigma * 4 * 2 + 1, low_sigma * 4 * 2 + 1)
    ls = cv2.GaussianBlur(fimg, ks, sigmaX=low_sigma, dst=None, sigmaY=low_sigma)
    hs = cv2.GaussianBlur(fimg, ks, sigmaX=high_sigma, dst=None, sigmaY=high_sigma)
    dog = hs - ls
    return cv2.normalize(dog, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U), fimg, ls, hs


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build(cv2):
    out, cases = {}, {}
    dtypes = {"u8": np.uint8, "u16": np.uint16, "f32": np.float32}

    # -- Farneback, small images: full flows ----------------------------------------------------------------------------
    ref, mov = make_pair(cv2, 160, 184, 3)
    for dn in ("u8", "f32"):
        r, m = to_dtype(ref, dtypes[dn]), to_dtype(mov, dtypes[dn])
        out[f"fb_{dn}_ref"], out[f"fb_{dn}_mov"] = r, m
        for win in (19, 99):
            for iters in (1, 3):
                name = f"fb_{dn}_w{win}_i{iters}"
                out[name] = cv_farneback(cv2, m, r, win, iters)
                cases[name] = dict(kind="farneback", ref=f"fb_{dn}_ref", mov=f"fb_{dn}_mov", winsize=win, iterations=iters)
    # -- Farneback, ONE 1200^2 window (tile 1000 + 2 x overlap 100, winsize 99, 3 iterations: the reference's defaults):
    #    stride-5 sample of the flow + SHA-256 of all of it; the inputs derive exactly (integer arithmetic) from a stored
    #    300^2 seed pair: upsample4, and frac_f32 for the float32 variant
    ref, mov = make_pair(cv2, 300, 300, 4)
    out["fbwin_seed_ref"], out["fbwin_seed_mov"] = to_dtype(ref, np.uint8), to_dtype(mov, np.uint8)
    r8, m8 = upsample4(out["fbwin_seed_ref"]), upsample4(out["fbwin_seed_mov"])
    for dn, (r, m) in (("u8", (r8, m8)), ("f32", (frac_f32(r8), frac_f32(m8)))):
        flow = cv_farneback(cv2, m, r, 99, 3)
        name = f"fbwin_{dn}_w99_i3"
        out[name + "_s5"] = np.ascontiguousarray(flow[::5, ::5])
        cases[name] = dict(kind="farneback_window", ref="fbwin_seed_ref", mov="fbwin_seed_mov", derive=dn, winsize=99,
                           iterations=3, sample=name + "_s5", stride=5, sha256=sha(flow), shape=list(flow.shape))

    # -- remap ---------------------------------------------------------------------------------------------------------
    base = texture(cv2, 151, 203, 5)
    out["remap_map"] = random_map(140, 180, 151, 203, 6)
    for dn, cn in (("u8", 1), ("u16", 1), ("f32", 1), ("f32", 2)):
        src = to_dtype(base, dtypes[dn])
        if cn == 2:
            src = np.ascontiguousarray(np.stack([base - 100.0, 60.0 - base * 0.5], -1).astype(np.float32))   # flows have both signs
        name = f"remap_{dn}_c{cn}"
        out[name + "_src"] = src
        out[name] = cv2.remap(src, out["remap_map"], None, cv2.INTER_LINEAR)
        cases[name] = dict(kind="remap", src=name + "_src", map="remap_map")

    # -- pyramids --------------------------------------------------------------------------------------------------------
    for shape in ((160, 200), (161, 203)):
        img = texture(cv2, shape[0], shape[1], 7)
        for dn in dtypes:
            name = f"pyrdown_{dn}_{shape[0]}x{shape[1]}"
            out[name + "_src"] = to_dtype(img, dtypes[dn])
            out[name] = cv2.pyrDown(out[name + "_src"])
            cases[name] = dict(kind="pyr_down", src=name + "_src")
    rs = np.random.RandomState(8)
    for src_shape, dst_hw in (((60, 80), (120, 160)), ((61, 82), (121, 163)), ((61, 82), (122, 164)), ((5, 7), (9, 13))):
        flow = rs.normal(0, 3, src_shape + (2,)).astype(np.float32)
        src_name = f"pyrup_{src_shape[0]}x{src_shape[1]}_to_{dst_hw[0]}x{dst_hw[1]}_src"
        out[src_name] = flow
        for scale in (1, 2, 4):                         # the reference multiplies by 2 (:140,150,164), 4 (:169) or nothing (:212)
            name = f"pyrup_{src_shape[0]}x{src_shape[1]}_to_{dst_hw[0]}x{dst_hw[1]}_x{scale}"
            out[name] = cv2.pyrUp(flow * scale, dstsize=dst_hw[::-1])
            cases[name] = dict(kind="pyr_up", src=src_name, scale=scale, dst_hw=list(dst_hw))

    # -- dog() chain, normalize -> u8 --------------------------------------------------------------------------------------
    img = texture(cv2, 200, 232, 9)
    for dn in dtypes:
        src = to_dtype(img, dtypes[dn])
        dog, fimg, ls, hs = cv_dog(cv2, src)
        name = f"dog_{dn}"
        out[name + "_src"], out[name] = src, dog
        cases[name] = dict(kind="dog", src=name + "_src", low_sigma=5, high_sigma=9)
        if dn == "u8":     # the intermediates of one chain: which step a build diverges at, if it does
            out[name + "_norm"], out[name + "_blur_lo"], out[name + "_blur_hi"] = fimg, ls, hs
            cases[name].update(norm=name + "_norm", blur_lo=name + "_blur_lo", blur_hi=name + "_blur_hi")
        name = f"normalize_u8_{dn}"
        out[name + "_src"] = src
        out[name] = cv2.normalize(src, None, 0, 255, cv2.NORM_MINMAX, cv2.CV_8U)
        cases[name] = dict(kind="normalize_u8", src=name + "_src")

    # -- warpAffine -------------------------------------------------------------------------------------------------------
    img = texture(cv2, 160, 200, 11)
    M = np.array([[0.998, -0.021, 4.3], [0.019, 1.003, -2.7]])
    out["affine_M"] = M
    for dn in dtypes:
        name = f"warp_affine_{dn}"
        out[name + "_src"] = to_dtype(img, dtypes[dn])
        out[name] = cv2.warpAffine(out[name + "_src"], M, dsize=(200, 160))
        cases[name] = dict(kind="warp_affine", src=name + "_src", M="affine_M", dsize=[200, 160])
    return out, cases


def build_facts(cv2):
    info = cv2.getBuildInformation()
    keep = [ln.strip() for ln in info.splitlines()
            if any(k in ln for k in ("Version control", "CPU/HW features", "Baseline", "Dispatched", "requested", "SSE", "AVX",
                                     "Intel IPP", "IPP", "Parallel framework", "OpenCL", "Lapack", "Platform", "Host:"))]
    hw = {k: bool(cv2.checkHardwareSupport(getattr(cv2, "CPU_" + k))) for k in ("SSE2", "SSE4_1", "AVX", "AVX2", "FMA3")
          if hasattr(cv2, "CPU_" + k)}
    return {"cv2_version": cv2.__version__, "numpy_version": np.__version__, "build_lines": keep, "cpu_features_in_use": hw,
            "ipp": bool(getattr(cv2, "ipp", None) and cv2.ipp.useIPP()) if hasattr(cv2, "ipp") else None,
            "threads": cv2.getNumThreads()}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default=None, help="output file (default: tests/golden/cv2_<version>.npz next to this script)")
    ap.add_argument("--any-version", action="store_true",
                    help=f"record with a cv2 other than {PINNED} (the file name and the meta data carry the version; "
                         "the consumers report it)")
    args = ap.parse_args()
    import cv2
    if not cv2.__version__.startswith(PINNED) and not args.any_version:
        sys.exit(f"cv2 {cv2.__version__} is not the reference's pin ({PINNED}, opencv-contrib-python=={PINNED}.64, "
                 "environment.yaml:75); pass --any-version to record it all the same")
    out, cases = build(cv2)
    meta = dict(format=FORMAT, facts=build_facts(cv2), cases=cases,
                standin=bool(getattr(cv2, "__microaligner_standin__", False)))
    version = ".".join(cv2.__version__.split(".")[:3])
    path = args.out or os.path.join(HERE, f"cv2_{version}.npz")
    np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), np.uint8), **out)
    print(f"wrote {path}: {len(cases)} cases, {os.path.getsize(path) / 1e6:.1f} MB, cv2 {cv2.__version__}")
    for ln in meta["facts"]["build_lines"]:
        print("   ", ln)


if __name__ == "__main__":
    main()
