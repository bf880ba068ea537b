"""Page-warp driver throughput (host pages in, host pages out): 8 u16 pages of 16384^2 with one flow."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd import Warper
from microaligner_amd.device import get_context
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
n = 8
rng = np.random.default_rng(0)
page = rng.integers(0, 65535, (H, W), dtype=np.uint16)
pages = [page] * n
flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 3.3; flow[..., 1] = -2.1
ctx = get_context()
w = Warper(); w.flow = ctx.asdevice(flow)
out = [np.empty_like(page) for _ in range(n)]
w.warp_pages(pages[:2], out[:2])
t0 = time.perf_counter(); w.warp_pages(pages, out); dt = time.perf_counter() - t0
print(f"warp_pages: {n} u16 pages {H}x{W}: {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s, {2*n*page.nbytes/dt/1e9:.1f} GB/s host<->device")
t0 = time.perf_counter()
for i in range(n):
    w2 = Warper(); w2.image, w2.flow = pages[i], w.flow; o = w2.warp()
dt = time.perf_counter() - t0
print(f"per-page warp(): {dt*1e3:.0f} ms = {n*H*W/dt/1e6:.0f} Mpix/s")
