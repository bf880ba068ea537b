"""Throughput of L concurrent lanes (one Context = one HIP stream + workspace per lane, one thread per lane) on one
device: python tools/lanes_probe.py [size] [steps_per_lane]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from microaligner_amd import OptFlowRegistrator, Warper, synthetic
from microaligner_amd.device import Context, use_context

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ref, mov = synthetic.make_pair(N, N, 1)
PARAMS = dict(num_pyr_lvl=4, num_iterations=3, tile_size=1000, overlap=100, use_full_res_img=True, use_dog=True)


def one(ctx, dref, dmov):
    reg = OptFlowRegistrator(); reg.verbose = False
    for k, v in PARAMS.items(): setattr(reg, k, v)
    reg.ref_img, reg.mov_img = dref, dmov
    flow = reg.register()
    w = Warper(); w.tile_size, w.overlap = 1000, 100
    w.image, w.flow = dmov, flow
    return w.warp()


def lane(ctx, steps, barrier, out, delay=0.0):
    with use_context(ctx):
        dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
        one(ctx, dref, dmov); ctx.sync()     # warm-up
        barrier.wait()
        t0 = time.perf_counter()
        time.sleep(delay)
        for _ in range(steps):
            r = one(ctx, dref, dmov)
        ctx.sync()
        out.append((t0, time.perf_counter()))


STAG = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
for L in (1, 2, 3):
    ctxs = [Context(0) for _ in range(L)]
    bar, out = threading.Barrier(L), []
    th = [threading.Thread(target=lane, args=(c, K, bar, out, i * STAG / L)) for i, c in enumerate(ctxs)]
    [t.start() for t in th]; [t.join() for t in th]
    t = max(b for a, b in out) - min(a for a, b in out)
    print(f"lanes={L}: {L * K} steps in {t * 1e3:.1f} ms -> {t * 1e3 / (L * K):.2f} ms/step, {N * N * L * K / t / 1e6:.0f} Mpix/s", flush=True)
    for c in ctxs: c.close()
