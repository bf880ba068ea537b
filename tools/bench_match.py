"""One 2-NN match of the feature stage's top level (2048^2 image of a 4096^2 mosaic tile, ~22 k x 22 k descriptors of 200 floats):
time per ma_knn2_l2 call, queries that needed the exact fallback.  python3 tools/bench_match.py [edge] [reps]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import synthetic                                       # noqa: E402
from microaligner_amd.device import get_context                              # noqa: E402
from microaligner_amd.feature_reg import feature_detection as FD             # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = get_context()
ref, mov, M = synthetic.make_mosaic_tile(2 * edge, 2 * edge, seed=1, dtype=np.float32)
fr = FD.find_features_of_device_image(ctx.dog_u8(ctx.pyr_down(ctx.asdevice(ref)), 5, 9), 1000, ctx)
fm = FD.find_features_of_device_image(ctx.dog_u8(ctx.pyr_down(ctx.asdevice(mov)), 5, 9), 1000, ctx)
q, t = fm.descriptors_for_search, fr.descriptors_for_search
for mode in ("filtered", "exact"):
    st = {}
    ctx.knn2(q, t, mode=mode, stats=st, on_device=True)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.knn2(q, t, mode=mode, on_device=True)
    ctx.sync()
    print(f"{mode}: nq {len(q)} nt {len(t)}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per search, uncertified {st.get('uncertified')}")
d = q.numpy()
n = np.sqrt((d.astype(np.float64) ** 2).sum(1))
print("descriptor norms: min %.4g median %.4g max %.4g; max |value| %.4g; share of |value| < 6.1e-5 (f16 normal range): %.4f"
      % (n.min(), np.median(n), n.max(), np.abs(d).max(), (np.abs(d) < 6.1e-5).mean()))
