"""One page through Warper.warp() (banded driver) with the driver's timeline (MA_TRACE_PAGES=1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd import Warper
from microaligner_amd.device import get_context, bind_to_device_numa
H = W = 16384
if "--bind" in sys.argv:
    print("bound to", len(bind_to_device_numa(0)), "cpus", file=sys.stderr)
rng = np.random.default_rng(0)
page = rng.integers(0, 65535, (H, W), dtype=np.uint16)
pages = [page ^ np.uint16(k) for k in range(4)]
flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 3.3; flow[..., 1] = -2.1
ctx = get_context()
dflow = ctx.asdevice(flow)
for i in range(4):
    t0 = time.perf_counter()
    w2 = Warper(); w2.image, w2.flow = pages[i], dflow; o = w2.warp()
    print(f"warp() page {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", file=sys.stderr)
