"""numpy-in / numpy-out anatomy of register() + warp() at 16384^2 (H2D of the images, D2H of flow and warped image):
where the host-inclusive milliseconds go.  Run on the GPU box: python tools/pcie_rate.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microaligner_amd import OptFlowRegistrator, Warper, synthetic  # noqa: E402
from microaligner_amd.device import get_context  # noqa: E402


def t(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        best = min(best, time.perf_counter() - t0)
    print(f"  {label}: {best * 1e3:.1f} ms")
    return r


ctx = get_context()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
params = dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True)
ref, mov = synthetic.make_pair(size, size, 1)
gb = ref.nbytes / 1e9
print(f"{size}x{size}: image {gb:.2f} GB, flow {2 * gb:.2f} GB")
d = t("H2D image, pageable numpy (asdevice)", lambda: ctx.asdevice(ref))
dflow = ctx.zeros((size, size, 2), np.float32)
ctx.sync()
t("D2H flow into pooled page-locked array (numpy())", lambda: dflow.numpy())
t("D2H flow into fresh np.empty (out=)", lambda: dflow.numpy(out=np.empty((size, size, 2), np.float32)), reps=2)
hflow = dflow.numpy()
t("H2D flow from page-locked array", lambda: ctx.asdevice(hflow))
pf = np.empty((size, size, 2), np.float32)
pf[:] = 0
t("H2D flow from pageable array", lambda: ctx.asdevice(pf))
del pf, hflow, dflow, d

reg = OptFlowRegistrator()
reg.verbose = False
for k, v in params.items():
    setattr(reg, k, v)
w = Warper()


def reg_only():
    reg.ref_img, reg.mov_img = ref, mov
    return reg.register()


def both():
    flow = reg_only()
    w.image, w.flow = mov, flow
    return w.warp()


both()
flow = t("register() numpy->numpy", reg_only)


def warp_only():
    w.image, w.flow = mov, flow
    return w.warp()


t("warp() numpy->numpy", warp_only)
t("register()+warp() numpy->numpy", both)
dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)


def dev():
    reg.ref_img, reg.mov_img = dref, dmov
    f = reg.register()
    w.image, w.flow = dmov, f
    r = w.warp()
    ctx.sync()
    return r


t("register()+warp() device-resident", dev)
