"""Soak run (not part of the test suite): N random configurations of register() + warp() -- shape, dtype, tile, overlap,
levels, DOG, rounding models, related / unrelated pairs -- through the C entry point against the oracle orchestration,
bit for bit.  python3 tools/soak_parity.py [N] [first seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import OptFlowRegistrator, Warper, synthetic   # noqa: E402
from oracle import oracle as O                                        # noqa: E402
from oracle import register_oracle as RO                              # noqa: E402

n, s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(230, SMAX)), int(rng.integers(230, SMAX))
    dtype = [np.uint8, np.uint16, np.float32][rng.integers(0, 3)]
    tile = int(rng.integers(100, 700))
    ov = int(rng.integers(10, min(tile // 2 - 1, 110)))
    p = dict(num_pyr_lvl=int(rng.integers(0, 4)), use_full_res_img=bool(rng.integers(0, 2)), use_dog=bool(rng.integers(0, 2)),
             tile_size=tile, overlap=ov, num_iterations=int(rng.integers(1, 4)))
    if p["num_pyr_lvl"] == 0:
        p["use_full_res_img"] = True
    if min(H, W) / 2 < 100 and not p["use_full_res_img"]:
        p["use_full_res_img"] = True
    fb_fused, dog_fused = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    make = synthetic.make_unrelated_pair if rng.integers(0, 4) == 0 else synthetic.make_pair
    ref, mov = make(H, W, seed, dtype)
    reg = OptFlowRegistrator()
    reg.verbose = False
    for k, v in p.items():
        setattr(reg, k, v)
    reg.muladd_fused, reg.dog_muladd_fused = fb_fused, dog_fused
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    w = Warper()
    w.tile_size, w.overlap = tile, ov
    w.image, w.flow = mov, flow
    warped = w.warp()
    exp, rep = RO.register(ref, mov, fused=fb_fused, dog_flags=O.DOG_FUSED if dog_fused else 0, nthreads=64, **p)
    ok = (np.array_equal(flow, exp) and np.array_equal(warped, RO.warp(mov, exp, tile, ov))
          and [r.accepted for r in reg.level_reports] == [r[3] for r in rep])
    bad += not ok
    print(seed, (H, W), np.dtype(dtype).name, p, "fma" if fb_fused else "", "dogfma" if dog_fused else "",
          [int(r.accepted) for r in reg.level_reports], "OK" if ok else "MISMATCH", flush=True)
print(f"{n} configurations, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
