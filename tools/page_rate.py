"""Page-warp driver throughput (host pages in, host pages out): 8 u16 pages of 16384^2 with one flow."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd import Warper
from microaligner_amd.device import get_context, bind_to_device_numa
if "--no-bind" not in sys.argv:
    bind_to_device_numa(0)      # as bench.py and the pipeline's ranks do: host buffers and copy threads next to the GPU
H = W = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16384
n = 8
rng = np.random.default_rng(0)
page = rng.integers(0, 65535, (H, W), dtype=np.uint16)
pages = [page ^ np.uint16(k) for k in range(n)]   # distinct arrays: nothing is recognised as resident
flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 3.3; flow[..., 1] = -2.1
ctx = get_context()
w = Warper(); w.flow = ctx.asdevice(flow)
out = [np.zeros_like(page) for _ in range(n)]
for o in out:
    o.fill(1)           # every page of the results exists before the clock starts
w.warp_pages(pages[:3], out[:3])      # three device slots, staging rings, copy threads exist from here on
t0 = time.perf_counter(); w.warp_pages(pages, out); dt = time.perf_counter() - t0
print(f"warp_pages, pageable results: {n} u16 pages {H}x{W}: {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s, {2*n*page.nbytes/dt/1e9:.1f} GB/s host<->device")
pout = [ctx.host_empty((H, W), np.uint16, limit=n) for _ in range(n)]
w.warp_pages(pages[:3], pout[:3])
t0 = time.perf_counter(); w.warp_pages(pages, pout); dt = time.perf_counter() - t0
print(f"warp_pages, page-locked results: {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s")
assert all(np.array_equal(a, b) for a, b in zip(out, pout))
del pout
def loop(k):
    for i in range(k):
        w2 = Warper(); w2.image, w2.flow = pages[i], w.flow; o = w2.warp()
    return o
loop(3)   # page-locked result buffers of the pool exist from here on
t0 = time.perf_counter()
o = loop(n)
dt = time.perf_counter() - t0
print(f"per-page warp() (banded driver): {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s")
assert np.array_equal(o, out[n - 1])
Warper.HOST_BANDED_MIN = 1 << 60
loop(3)
t0 = time.perf_counter()
o = loop(n)
dt = time.perf_counter() - t0
print(f"per-page warp() (upload, warp, download): {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s")
assert np.array_equal(o, out[n - 1])
