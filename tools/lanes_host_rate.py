"""numpy-in / numpy-out throughput with several pairs in flight (parallel.register_pairs(lanes=...)): transfers of one
pair overlap the kernels of another.  python tools/lanes_host_rate.py [size] [pairs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microaligner_amd import parallel, synthetic  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
params = dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True)
ref, mov = synthetic.make_pair(size, size, 1)
pairs = [(ref, mov)] * npairs
for lanes in (1, 2, 3):
    parallel.register_pairs(pairs[:lanes], params, warp=True, lanes=lanes)          # warm pools / contexts
    t0 = time.perf_counter()
    out = parallel.register_pairs(pairs, params, warp=True, lanes=lanes)
    dt = (time.perf_counter() - t0) / npairs
    print(f"lanes {lanes}: {dt * 1e3:.1f} ms per pair numpy -> numpy = {size * size / dt / 1e6:.0f} Mpix/s", flush=True)
    del out
