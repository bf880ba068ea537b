"""Host-thread scaling of the CPU oracle's stages (bench.py's cpu_baseline leg) on this box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from microaligner_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import register_oracle as RO  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ref, mov = synthetic.make_pair(size, size, 1)
params = dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True)
for nt in [int(a) for a in sys.argv[2:]] or [16, 64, 128, os.cpu_count()]:
    O.set_threads(nt)
    t0 = time.perf_counter(); O.dog(ref); t1 = time.perf_counter(); O.dog(ref); t2 = time.perf_counter()
    O.pyr_down(ref); t3 = time.perf_counter()
    st = {}
    t4 = time.perf_counter()
    RO.register(ref, mov, nthreads=nt, stage_seconds=st, **params)
    t5 = time.perf_counter()
    print(f"threads {nt}: dog {t1 - t0:.3f} / {t2 - t1:.3f} s, pyr_down {t3 - t2:.3f} s, register {t5 - t4:.1f} s",
          {k: round(v, 2) for k, v in st.items()}, flush=True)
