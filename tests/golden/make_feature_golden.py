"""Generates tests/golden/feature_index.json by running the REFERENCE's own FeatureRegistrator.

Build container only (needs /root/reference).  As in make_golden.py nothing of the reference is copied: its class
is imported and driven, with `cv2` / `dask` / `sklearn` / `skimage` replaced by stand-ins.  The dense calls
(pyrDown, normalize, GaussianBlur, warpAffine, NMI) forward to the C oracle; the sparse calls
(FastFeatureDetector, xfeatures2d.DAISY, FlannBasedMatcher.knnMatch, estimateAffinePartial2D) forward to
microaligner_amd/feature_reg/sparse_cpu.py -- the same functions the product uses -- so what the fixture pins is
the reference's ORCHESTRATION (feature_reg/feature_registrator.py:70-312, tile_registration.py,
feature_detection.py:88-158): pyramid order, per-tile limits and coordinate bookkeeping, the iteration loop with
its mutual-information and plausibility gates, and the matrix algebra.  It does not pin opencv-contrib's own
FAST/DAISY/FLANN/RANSAC arithmetic (absent here; see sparse_cpu.py).

    python tests/golden/make_feature_golden.py
"""
import contextlib
import io
import json
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import oracle as O  # noqa: E402
from microaligner_amd import synthetic  # noqa: E402
from microaligner_amd.feature_reg import sparse_cpu as SP  # noqa: E402
import make_golden  # noqa: E402  (installs the dense stand-ins)


def _install_sparse_standins():
    cv2 = sys.modules["cv2"]

    class KeyPoint:
        def __init__(self, x=0.0, y=0.0, size=7.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
            self.pt, self.size, self.angle = (x, y), size, angle
            self.response, self.octave, self.class_id = response, octave, class_id

    cv2.KeyPoint = KeyPoint
    cv2.FAST_FEATURE_DETECTOR_TYPE_9_16 = 2
    cv2.RANSAC = 8

    class _Fast:
        def __init__(self, threshold, nonmax):
            self.t, self.nm = threshold, nonmax

        def detect(self, img):
            return [KeyPoint(k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id)
                    for k in SP.fast_detect(np.ascontiguousarray(img), self.t, self.nm)]

    def FastFeatureDetector_create(threshold=10, nonmaxSuppression=True, type=2):
        assert type == 2
        return _Fast(threshold, nonmaxSuppression)

    class _Daisy:
        def __init__(self, **kw):
            self.d = SP.Daisy(kw["radius"], kw["q_radius"], kw["q_theta"], kw["q_hist"])

        def compute(self, img, kp):
            return kp, self.d.compute(img, [SP.KeyPoint(k.pt, k.size, k.angle, k.response, k.octave, k.class_id)
                                            for k in kp])

    xf = types.SimpleNamespace(DAISY_NRM_NONE=100)

    def DAISY_create(radius=15, q_radius=3, q_theta=8, q_hist=8, norm=100, interpolation=True, use_orientation=False):
        assert norm == 100 and interpolation and not use_orientation
        return _Daisy(radius=radius, q_radius=q_radius, q_theta=q_theta, q_hist=q_hist)

    xf.DAISY_create = DAISY_create
    cv2.xfeatures2d = xf

    class _DMatch:
        def __init__(self, q, t, d):
            self.queryIdx, self.trainIdx, self.distance = q, t, d

    class _Matcher:
        def knnMatch(self, query, train, k=2):
            assert k == 2
            idx, dist = SP.knn2_sequential(query, train)   # the definition the device search implements, bit for bit
            return [[_DMatch(q, int(idx[q, 0]), dist[q, 0]), _DMatch(q, int(idx[q, 1]), dist[q, 1])]
                    for q in range(len(idx))]

    def estimateAffinePartial2D(from_pts, to_pts, method=8, confidence=0.99):
        assert method == 8
        M, mask = SP.estimate_affine_partial_2d(from_pts, to_pts, confidence=confidence)
        return M, mask

    def warpAffine(img, M, dsize=None):
        return O.warp_affine(img, M, dsize)

    cv2.FastFeatureDetector_create = FastFeatureDetector_create
    cv2.FlannBasedMatcher_create = lambda: _Matcher()
    cv2.estimateAffinePartial2D = estimateAffinePartial2D
    cv2.warpAffine = warpAffine


def affine(theta_deg, scale, tx, ty):
    th = np.deg2rad(theta_deg)
    return np.array([[scale * np.cos(th), -scale * np.sin(th), tx], [scale * np.sin(th), scale * np.cos(th), ty]])


CASES = {
    # two pyramid levels (factors 4, 2), 3 x 3 .. 2 x 2 tiles, small rotation + shift: accepted everywhere
    "cells_900x1000_rot": dict(shape=(900, 1000), seed=11, M=[0.6, 1.0, 9.0, -6.0],
                               params=dict(num_pyr_lvl=2, tile_size=200)),
    # with the full-resolution level and a slight scale change
    "cells_640x720_full": dict(shape=(640, 720), seed=12, M=[-0.4, 1.004, -5.0, 7.0],
                               params=dict(num_pyr_lvl=1, use_full_res_img=True, tile_size=300, num_iterations=2)),
    # unrelated images: nothing matches or the gate rejects -> identity
    "cells_unrelated": dict(shape=(800, 800), seed=13, M=None, params=dict(num_pyr_lvl=2, tile_size=200)),
    # without DOG on the feature images (use_dog=False), uint8 inputs
    "cells_u8_nodog": dict(shape=(700, 760), seed=14, M=[0.0, 1.0, 6.0, 4.0], dtype="uint8",
                           params=dict(num_pyr_lvl=1, tile_size=250, use_dog=False)),
}


def make_inputs(case):
    H, W = case["shape"]
    dt = np.dtype(case.get("dtype", "uint16"))
    ref = synthetic.make_cells(H, W, seed=case["seed"], dtype=dt)
    if case["M"] is None:
        return ref, synthetic.make_cells(H, W, seed=case["seed"] + 1000, dtype=dt)
    return ref, O.warp_affine(ref, affine(*case["M"]))


def main():
    make_golden._install_standins()
    _install_sparse_standins()
    sys.path.insert(0, "/root/reference")
    from microaligner import FeatureRegistrator  # the reference's own class

    index = {}
    for name, case in CASES.items():
        ref, mov = make_inputs(case)
        freg = FeatureRegistrator()
        for k, v in case["params"].items():
            setattr(freg, k, v)
        freg.ref_img, freg.mov_img = ref, mov
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T = freg.register()
        log = buf.getvalue()
        mi = [(float(a), float(b)) for a, b in re.findall(r"MI score after: (\S+) \| MI score before: (\S+)", log)]
        good = [[int(a), int(b)] for a, b in re.findall(r"Good matches (\d+) / (\d+)", log)]
        accepted = [("Better" in ln) for ln in log.splitlines() if "alignment than before" in ln]
        factors = [int(f) for f in re.findall(r"Pyramid factor (\d+)", log)]
        index[name] = dict(shape=list(case["shape"]), seed=case["seed"], M=case["M"], dtype=case.get("dtype", "uint16"),
                           params=case["params"], factors=factors, mi=mi, good_matches=good, accepted=accepted,
                           t_mat=[[float(v) for v in row] for row in np.asarray(T)])
        print(name, "factors", factors, "accepted", accepted, "\n", np.asarray(T))
    with open(os.path.join(HERE, "feature_index.json"), "w") as f:
        json.dump(index, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
