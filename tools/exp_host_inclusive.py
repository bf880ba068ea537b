"""Where the numpy-in / numpy-out pass of cfg3 spends its time: phase timers around the drop-in API."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from microaligner_amd import OptFlowRegistrator, Warper
from microaligner_amd.device import get_context
from microaligner_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
ctx = get_context()
ref, mov = synthetic.make_pair(n, n, seed=1, dtype=np.float32)
reg = OptFlowRegistrator(); reg.verbose = False
reg.use_dog = True; reg.num_pyr_lvl = 5
w = Warper(); w.tile_size, w.overlap = reg.tile_size, reg.overlap

def T(label, fn, acc):
    t0 = time.perf_counter(); r = fn(); ctx.sync(); acc.setdefault(label, []).append((time.perf_counter() - t0) * 1e3); return r

for rep in range(4):
    acc = {}
    ctx.forget_host_arrays()
    t_all = time.perf_counter()
    dref = T("asdevice ref", lambda: ctx.asdevice(ref), acc)
    dmov = T("asdevice mov", lambda: ctx.asdevice(mov), acc)
    reg.ref_img, reg.mov_img = ref, mov
    flow = T("register (arrays resident, flow to host)", reg.register, acc)
    w.image, w.flow = mov, flow
    warped = T("warp", w.warp, acc)
    total = (time.perf_counter() - t_all) * 1e3
    dflow = ctx.asdevice(flow)
    t0 = time.perf_counter(); h = dflow.numpy(); d2h = (time.perf_counter() - t0) * 1e3
    del h
    print(f"pass {rep}: total {total:.1f} ms | " + " | ".join(f"{k} {v[0]:.1f}" for k, v in acc.items()) + f" | (flow D2H alone {d2h:.1f})", flush=True)
print(ctx.transfer_stats())
