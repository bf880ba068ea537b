"""Time of ma_feature_extract (FAST + selection + DAISY) on one level image: python3 tools/bench_extract.py [edge] [reps]
MICROALIGNER_HIP_LIB selects another build (tools/build_variant.py)."""
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
ref, mov, M = synthetic.make_mosaic_tile(edge, edge, seed=1, dtype=np.float32)
img = ctx.dog_u8(ctx.asdevice(ref), 5, 9)
f = FD.find_features_of_device_image(img, 1000, ctx)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    f = FD.find_features_of_device_image(img, 1000, ctx)
ctx.sync()
print(f"edge {edge}: {1e3 * (time.perf_counter() - t0) / reps:.3f} ms per extraction, {len(f.descriptors_for_search)} keypoints")
