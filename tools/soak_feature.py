"""Soak (not part of the suite): FeatureRegistrator.register() with the feature stage on the device (ma_feature_extract,
ma_knn2_l2, ma_match_similarity) against the same registrator with features and matching computed by the HOST statement
(feature_reg/sparse_cpu.py: FAST, DAISY, the sequential exact 2-NN, the ratio test, RANSAC) -- N random configurations; matrix and
log must be identical.  python3 tools/soak_feature.py [N] [first seed]"""
import contextlib
import io
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import FeatureRegistrator, synthetic       # noqa: E402
from oracle import oracle as O                                   # noqa: E402 (test infrastructure: the warp that makes the moving image)

n, s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(420, 900)), int(rng.integers(420, 900))
    dtype = [np.uint8, np.uint16, np.float32][rng.integers(0, 3)]
    ref = synthetic.make_cells(H, W, seed=seed, dtype=dtype)
    if rng.integers(0, 5) == 0:
        mov = synthetic.make_cells(H, W, seed=seed + 5000, dtype=dtype)           # unrelated: rejections, few matches
    else:
        th, sc = np.deg2rad(rng.uniform(-1.0, 1.0)), rng.uniform(0.99, 1.01)
        M = np.array([[sc * np.cos(th), -sc * np.sin(th), rng.uniform(-12, 12)], [sc * np.sin(th), sc * np.cos(th), rng.uniform(-12, 12)]])
        if rng.integers(0, 3) == 0:
            M = np.array([[1.0, 0.0, float(rng.integers(-9, 9))], [0.0, 1.0, float(rng.integers(-9, 9))]])   # exact integer shift
        mov = O.warp_affine(ref, M)
    p = dict(num_pyr_lvl=int(rng.integers(0, 3)), num_iterations=int(rng.integers(1, 4)), tile_size=int(rng.integers(150, 500)),
             use_full_res_img=bool(rng.integers(0, 2)), use_dog=bool(rng.integers(0, 4) > 0 or dtype != np.uint8))
    if p["num_pyr_lvl"] == 0 or min(H, W) / 2 < 100:
        p["use_full_res_img"] = True
    out = []
    for host in (False, True):
        f = FeatureRegistrator()
        for k, v in p.items():
            setattr(f, k, v)
        f.features_on_host = host
        f.ref_img, f.mov_img = ref, mov
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            try:
                T = f.register()
            except ValueError as e:
                T = str(e)
        out.append((T, buf.getvalue()))
    same = type(out[0][0]) is type(out[1][0]) and np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
    bad += not same
    print(seed, (H, W), np.dtype(dtype).name, p, "OK" if same else "MISMATCH", flush=True)
    if not same:
        print(out[0][0], out[1][0], sep="\n")
        a, b = out[0][1].splitlines(), out[1][1].splitlines()
        for x, y in zip(a, b):
            if x != y:
                print("  device:", x, "\n  host:  ", y)
                break
print(f"{n} configurations, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
