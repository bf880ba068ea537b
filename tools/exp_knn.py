"""2-NN search of n x n descriptors of 200 floats: exact kernel against the filtered (matrix-core) search."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from microaligner_amd.device import get_context
ctx = get_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 45000
rng = np.random.default_rng(1)
a = rng.gamma(0.3, 1.0, (n, 200)).astype(np.float32); a /= np.sqrt((a * a).sum(1, keepdims=True))
t = a
q = (t[rng.permutation(n)] + 0.02 * rng.random((n, 200), dtype=np.float32)).astype(np.float32)
dq, dt = ctx.asdevice(q), ctx.asdevice(t)
res = {}
for mode in ("exact", "filtered", "exact", "filtered"):
    st = {}
    ctx.sync(); t0 = time.perf_counter()
    i, d = ctx.knn2(dq, dt, mode=mode, stats=st)
    ctx.sync(); dt_ms = (time.perf_counter() - t0) * 1e3
    res[mode] = (i, d)
    print(f"{mode}: {dt_ms:.1f} ms  uncertified {st}", flush=True)
print("identical:", np.array_equal(res["exact"][0], res["filtered"][0]) and np.array_equal(res["exact"][1], res["filtered"][1]))
