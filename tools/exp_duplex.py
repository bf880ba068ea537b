"""Transfer-engine rates, one direction alone and both at once, page-locked and pageable host memory (1 GiB each way)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd.device import get_context, bind_to_device_numa
if "--bind" in sys.argv:
    print("bound to", len(bind_to_device_numa(0)), "cpus")
ctx = get_context()
N = 1 << 30
d_in, d_out = ctx.empty((N,), np.uint8), ctx.empty((N,), np.uint8)
src_pg, dst_pg = np.ones(N, np.uint8), np.ones(N, np.uint8)
src_pl, dst_pl = ctx.host_empty((N,), np.uint8), ctx.host_empty((N,), np.uint8)
src_pl[:] = 1; dst_pl[:] = 1


def timed(fn, reps=4):
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def both(up, down):
    res = {}
    def run(name, fn):
        res[name] = timed(fn)
    ts = [threading.Thread(target=run, args=("up", up)), threading.Thread(target=run, args=("down", down))]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return res["up"], res["down"], (time.perf_counter() - t0) / 4


def heavy():
    """~0.5 s of shader-bound work on another context (what a register() before the page warps is)."""
    from microaligner_amd import OptFlowRegistrator, synthetic
    from microaligner_amd.device import Context, use_context
    r, m = synthetic.make_pair(4096, 4096, seed=1, dtype=np.float32)
    c = Context(0)
    with use_context(c):
        reg = OptFlowRegistrator(); reg.verbose = False
        reg.num_pyr_lvl, reg.use_full_res_img, reg.use_dog = 3, True, True
        reg.ref_img, reg.mov_img = c.asdevice(r), c.asdevice(m)
        for _ in range(6):
            reg.register()
        c.sync()
    c.close()


AFTER_WORK = "--after-work" in sys.argv
for label, src, dst in (("page-locked", src_pl, dst_pl), ("pageable", src_pg, dst_pg), ("up pageable, down page-locked", src_pg, dst_pl),
                        ("up page-locked, down pageable", src_pl, dst_pg)):
    up = lambda: ctx.engine_upload(d_in, src)
    down = lambda: ctx.engine_download(d_out, dst)
    up(); down()
    if AFTER_WORK:
        heavy()
        a = timed(up, 2); heavy(); b = timed(down, 2); heavy()
    else:
        a, b = timed(up), timed(down)
    u2, d2, wall = both(up, down)
    print(f"{label:32s} alone: up {N / a / 1e9:5.1f} GB/s, down {N / b / 1e9:5.1f} GB/s | together: up {N / u2 / 1e9:5.1f}, down "
          f"{N / d2 / 1e9:5.1f}, sum {2 * N / wall / 1e9:5.1f} GB/s")
