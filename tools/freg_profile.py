"""Where FeatureRegistrator.register() spends its time on one 4096^2 mosaic tile (BASELINE cfg5): cProfile by cumulative time."""
import cProfile
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import FeatureRegistrator, synthetic   # noqa: E402
from microaligner_amd.device import get_context               # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ref, mov, M = synthetic.make_mosaic_tile(H, H, seed=1, dtype=np.float32)
freg = FeatureRegistrator()
freg.verbose = False
freg.ref_img, freg.mov_img = ref, mov
freg.register()
get_context().sync()
t0 = time.perf_counter()
for _ in range(3):
    T = freg.register()
get_context().sync()
print("register():", (time.perf_counter() - t0) / 3 * 1e3, "ms")
pr = cProfile.Profile()
pr.enable()
freg.register()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
