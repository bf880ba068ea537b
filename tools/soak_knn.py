"""Soak (not part of the suite): the filtered 2-NN search (split-float16 shortlist + certificate) against the exact kernel on
N random problems -- sizes, descriptor lengths, data kinds incl. real DAISY descriptors of synthetic cell images at several
scales and of their transformed copies.  python3 tools/soak_knn.py [N]"""
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from microaligner_amd import synthetic                                       # noqa: E402
from microaligner_amd.device import get_context                              # noqa: E402
from microaligner_amd.feature_reg import feature_detection as FD             # noqa: E402
from oracle import oracle as O                                               # noqa: E402 (test infrastructure: the warp of the moving image)

ctx = get_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = unc = tot = 0
rng = np.random.default_rng(7)
for case in range(n):
    kind = ["daisy", "daisy", "uniform", "hist", "clusters", "wide"][case % 6]
    if kind == "daisy":
        H, W = int(rng.integers(500, 2600)), int(rng.integers(500, 2600))
        ref = synthetic.make_cells(H, W, seed=100 + case)
        th = np.deg2rad(rng.uniform(-1, 1))
        M = np.array([[np.cos(th), -np.sin(th), rng.uniform(-15, 15)], [np.sin(th), np.cos(th), rng.uniform(-15, 15)]])
        mov = O.warp_affine(ref, M)
        tile = int(rng.choice([300, 500, 1000]))
        fr = FD.find_features_of_device_image(ctx.dog_u8(ctx.asdevice(ref), 5, 9), tile, ctx)
        fm = FD.find_features_of_device_image(ctx.dog_u8(ctx.asdevice(mov), 5, 9), tile, ctx)
        q, t = fm.descriptors_for_search, fr.descriptors_for_search
    else:
        nq, nt = int(rng.integers(300, 9000)), int(rng.integers(600, 12000))
        dim = int(rng.choice([200, 200, 128, 64, 208, 36]))
        if kind == "uniform":
            qh, th_ = rng.random((nq, dim)), rng.random((nt, dim))
        elif kind == "hist":
            th_ = rng.gamma(0.3, 1.0, (nt, dim)); th_ /= np.sqrt((th_ * th_).sum(1, keepdims=True))
            qh = th_[rng.integers(0, nt, nq)] + rng.uniform(0.001, 0.05) * rng.gamma(0.3, 1.0, (nq, dim))
        elif kind == "clusters":
            c = rng.random((50, dim))
            s = 10.0 ** rng.uniform(-5, -2)
            th_ = c[rng.integers(0, 50, nt)] + s * rng.standard_normal((nt, dim))
            qh = c[rng.integers(0, 50, nq)] + s * rng.standard_normal((nq, dim))
        else:
            mag = np.exp(rng.uniform(np.log(1e-5), np.log(1e2), (nt, dim)))
            th_ = rng.standard_normal((nt, dim)) * mag
            qh = th_[rng.integers(0, nt, nq)] * (1 + 1e-3 * rng.standard_normal((nq, dim)))
        scale = 10.0 ** rng.uniform(-6, 6)
        q, t = ctx.asdevice((qh * scale).astype(np.float32)), ctx.asdevice((th_ * scale).astype(np.float32))
    st = {}
    fi, fd = ctx.knn2(q, t, mode="filtered", stats=st)
    ei, ed = ctx.knn2(q, t, mode="exact")
    ok = np.array_equal(fi, ei) and np.array_equal(fd, ed)
    bad += not ok
    unc += st["uncertified"]; tot += len(fi)
    print(case, kind, len(fi), "x", t.shape[0], "dim", t.shape[1], "uncertified", st["uncertified"], "OK" if ok else "MISMATCH", flush=True)
print(f"{n} problems, {bad} mismatches, {unc} of {tot} queries through the exact fallback")
sys.exit(1 if bad else 0)
