"""Randomised comparison of the filtered 2-NN search (matrix-core shortlist + certificate) with the exact kernel:
shapes, descriptor lengths and value distributions drawn at random, results compared bit for bit.
    python3 tools/soak_knn.py [cases] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from microaligner_amd.device import get_context

ctx = get_context()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
kinds = ["uniform", "normal", "big", "tiny", "noisy_copies", "integers", "mixed_norms", "sparse", "one_hot"]
bad = 0
unc_total = q_total = 0
t0 = time.time()
for c in range(cases):
    kind = kinds[rng.integers(len(kinds))]
    nq, nt = int(rng.integers(1, 3000)), int(rng.integers(2, 6000))
    dim = 4 * int(rng.integers(1, 55))
    if kind == "uniform":
        q, t = rng.random((nq, dim)), rng.random((nt, dim))
    elif kind == "normal":
        q, t = rng.standard_normal((nq, dim)), rng.standard_normal((nt, dim))
    elif kind == "big":
        q, t = 1e4 * rng.standard_normal((nq, dim)), 1e4 * rng.standard_normal((nt, dim))
    elif kind == "tiny":
        q, t = 1e-4 * rng.random((nq, dim)), 1e-4 * rng.random((nt, dim))
    elif kind == "noisy_copies":
        base = rng.random((int(rng.integers(1, 40)), dim))
        eps = 10.0 ** rng.uniform(-8, -2)
        t = base[rng.integers(0, len(base), nt)] + eps * rng.standard_normal((nt, dim))
        q = base[rng.integers(0, len(base), nq)] + eps * rng.standard_normal((nq, dim))
    elif kind == "integers":
        q, t = rng.integers(0, 4, (nq, dim)).astype(float), rng.integers(0, 4, (nt, dim)).astype(float)
    elif kind == "mixed_norms":
        q = rng.standard_normal((nq, dim)) * 10.0 ** rng.uniform(-3, 3, (nq, 1))
        t = rng.standard_normal((nt, dim)) * 10.0 ** rng.uniform(-3, 3, (nt, 1))
    elif kind == "sparse":
        q = rng.random((nq, dim)) * (rng.random((nq, dim)) < 0.1)
        t = rng.random((nt, dim)) * (rng.random((nt, dim)) < 0.1)
    else:
        q = np.eye(dim)[rng.integers(0, dim, nq)]
        t = np.eye(dim)[rng.integers(0, dim, nt)]
    q, t = q.astype(np.float32), t.astype(np.float32)
    dq, dt = ctx.asdevice(np.ascontiguousarray(q)), ctx.asdevice(np.ascontiguousarray(t))
    st = {}
    fi, fd = ctx.knn2(dq, dt, mode="filtered", stats=st)
    ei, ed = ctx.knn2(dq, dt, mode="exact")
    ok = np.array_equal(fi, ei) and np.array_equal(fd, ed, equal_nan=True)
    unc_total += st["uncertified"]; q_total += nq
    if not ok:
        bad += 1
        wrong = np.nonzero((fi != ei).any(1) | (fd != ed).any(1))[0]
        print(f"MISMATCH case {c} {kind} nq={nq} nt={nt} dim={dim} uncertified={st['uncertified']} rows {wrong[:5]} "
              f"filtered {fi[wrong[0]]} {fd[wrong[0]]} exact {ei[wrong[0]]} {ed[wrong[0]]}", flush=True)
    dq.free(); dt.free()
print(f"{cases} cases, {bad} mismatching, {unc_total}/{q_total} queries through the exact fallback, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
