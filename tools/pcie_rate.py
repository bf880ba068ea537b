"""numpy-in / numpy-out rate of register() + warp() (includes H2D of both images, D2H of flow and warped image)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microaligner_amd import OptFlowRegistrator, Warper, synthetic

for size, params in ((4096, dict(num_pyr_lvl=2, use_full_res_img=True)),
                     (16384, dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True))):
    ref, mov = synthetic.make_pair(size, size, 1)
    for rep in range(2):
        t0 = time.perf_counter()
        reg = OptFlowRegistrator(); reg.verbose = False
        for k, v in params.items(): setattr(reg, k, v)
        reg.ref_img, reg.mov_img = ref, mov
        flow = reg.register()
        w = Warper(); w.image, w.flow = mov, flow
        out = w.warp()
        dt = time.perf_counter() - t0
    print(f"{size}x{size} numpy->numpy register+warp: {dt*1e3:.1f} ms = {size*size/dt/1e6:.1f} Mpix/s")
