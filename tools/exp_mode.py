"""What puts the page-warp driver's duplex transfers into the slow mode (bench: after the lanes leg)?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from microaligner_amd import Warper, OptFlowRegistrator, synthetic
from microaligner_amd.device import get_context, bind_to_device_numa, Context, use_context
bind_to_device_numa(0)
H = W = 16384
n = 8
rng = np.random.default_rng(0)
page = rng.integers(0, 65535, (H, W), dtype=np.uint16)
pages = [page ^ np.uint16(k) for k in range(n)]
flow = np.zeros((H, W, 2), np.float32); flow[..., 0] = 3.3; flow[..., 1] = -2.1
ctx = get_context()
w = Warper(); w.flow = ctx.asdevice(flow)
out = [np.ones_like(page) for _ in range(n)]


def rate(label):
    w.warp_pages(pages[:3], out[:3])
    t0 = time.perf_counter(); w.warp_pages(pages, out); dt = time.perf_counter() - t0
    print(f"{label:60s} {dt*1e3:6.0f} ms = {n*H*W/dt/1e9:5.1f} Gpix/s", flush=True)


def lanes(k, work, close=True):
    r, m = synthetic.make_pair(4096, 4096, seed=1, dtype=np.float32)
    def lane():
        c = Context(0)
        with use_context(c):
            if work:
                reg = OptFlowRegistrator(); reg.verbose = False
                reg.num_pyr_lvl, reg.use_full_res_img, reg.use_dog = 3, True, True
                reg.ref_img, reg.mov_img = c.asdevice(r), c.asdevice(m)
                reg.register(); c.sync()
        if close:
            c.close()
        else:
            keep.append(c)
    th = [threading.Thread(target=lane) for _ in range(k)]
    [t.start() for t in th]; [t.join() for t in th]


keep = []
mode = sys.argv[1]
rate("baseline")
if mode == "create_close":
    lanes(3, False); rate("after 3 contexts created and closed (no work)")
elif mode == "work_close":
    lanes(3, True); rate("after 3 contexts ran a register() and were closed")
elif mode == "work_keep":
    lanes(3, True, close=False); rate("after 3 contexts ran a register(), still open")
elif mode == "one_work_close":
    lanes(1, True); rate("after 1 context ran a register() and was closed")
elif mode == "work_close_sleep":
    lanes(3, True); time.sleep(3.0); rate("after 3 contexts ran a register() and were closed, 3 s later")
elif mode == "work_close_series":
    lanes(3, True)
    for i in range(8):
        rate(f"after 3 contexts ran a register() and were closed, call {i}")
elif mode == "trim":
    lanes(3, True); ctx.trim(); w.flow = ctx.asdevice(flow); rate("after 3 contexts + ctx.trim() of the main context")
rate("again")
