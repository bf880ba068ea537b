"""FeatureRegistrator.register() on one 4096^2 tile with a device sync around every Context method: true time per call."""
import sys, time, collections
import numpy as np
sys.path.insert(0, ".")
from microaligner_amd import FeatureRegistrator, synthetic
from microaligner_amd.device import get_context, Context
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ref, mov, M = synthetic.make_mosaic_tile(H, H, seed=1, dtype=np.float32)
freg = FeatureRegistrator(); freg.verbose = False
freg.ref_img, freg.mov_img = ref, mov
freg.register()
ctx = get_context()
acc = collections.defaultdict(lambda: [0, 0.0, []])
def wrap(name):
    fn = getattr(Context, name)
    def w(self, *a, **k):
        self.sync(); t0 = time.perf_counter()
        st = {}
        if name == "knn2": k = dict(k, stats=st)
        r = fn(self, *a, **k)
        self.sync(); dt = (time.perf_counter() - t0) * 1e3
        e = acc[name]; e[0] += 1; e[1] += dt
        if name in ("daisy_describe", "fast_keypoints"):
            e[2].append((tuple(a[0].shape), round(dt, 2)))
        if name == "knn2":
            t1 = time.perf_counter(); fn(self, *a, mode="exact"); self.sync()
            e[2].append((a[0].shape[0], a[1].shape[0], round(dt, 2), st, "exact", round((time.perf_counter() - t1) * 1e3, 2)))
        return r
    setattr(Context, name, w)
for n in ("knn2", "fast_keypoints", "daisy_describe", "dog_u8", "cut_tiles", "nmi_scores", "warp_affine_cv", "asdevice", "pyr_down"):
    if hasattr(Context, n): wrap(n)
ctx.sync(); t0 = time.perf_counter()
freg.register()
ctx.sync(); print("register() with syncs:", round((time.perf_counter() - t0) * 1e3, 1), "ms")
for k, (n, ms, extra) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:18s} {n:4d} calls {ms:8.2f} ms", extra if extra else "")
