"""Soak of the page-warp driver: random shapes, tilings, dtypes, page counts, band sizes and buffer kinds against the
single-page device warp of the same inputs (itself bit-exact against the oracle in tests/).  python tools/soak_pages.py [seconds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd import _lib as L
from microaligner_amd.device import get_context

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ctx = get_context()
rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "1")))
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    H, W = int(rng.integers(40, 3200)), int(rng.integers(40, 2600))
    if rng.random() < 0.15:
        tile, ov = 0, 0
    else:
        tile = int(rng.integers(30, 1300))
        ov = int(rng.integers(0, tile + 1)) if rng.random() < 0.5 else int(rng.integers(0, max(1, tile // 4)))
    dtype = [np.uint8, np.uint16, np.float32][int(rng.integers(0, 3))]
    npages = int(rng.integers(1, 8))
    band = [1, 1 << 20, 8 << 20, 32 << 20][int(rng.integers(0, 4))]
    amp = float(rng.choice([0.0, 0.7, 5.0, 60.0, 400.0]))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    flow = np.stack([amp * np.sin(yy / 37.0 + xx / 91.0), amp * np.cos(xx / 53.0 - yy / 29.0)], -1).astype(np.float32)
    if rng.random() < 0.2:
        flow[int(rng.integers(0, H)), int(rng.integers(0, W))] = np.nan
    hi = 255 if dtype == np.uint8 else 65535
    pages, outs = [], []
    for k in range(npages):
        p = (rng.random((H, W)) * hi).astype(dtype)
        if rng.random() < 0.4:
            q = ctx.host_empty((H, W), dtype, limit=16); q[...] = p; p = q
        pages.append(p)
        outs.append(ctx.host_empty((H, W), dtype, limit=16) if rng.random() < 0.4 else np.empty((H, W), dtype))
    dflow = ctx.asdevice(flow)
    exp = [ctx.warp(ctx.asdevice(np.array(p)), dflow, tile, ov).numpy() for p in pages]
    ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, band)
    try:
        ctx.warp_pages(pages, dflow, tile, ov, outs)
    finally:
        ctx.set_option(L.MA_OPT_WARP_BAND_BYTES, 32 << 20)
    for k in range(npages):
        same = np.array_equal(outs[k], exp[k], equal_nan=True) if dtype == np.float32 else np.array_equal(outs[k], exp[k])
        if not same:
            bad += 1
            print("MISMATCH", dict(H=H, W=W, tile=tile, ov=ov, dtype=np.dtype(dtype).name, npages=npages, band=band, amp=amp, page=k), flush=True)
    n += 1
print(f"{n} configurations, {bad} mismatching pages")
sys.exit(1 if bad else 0)
