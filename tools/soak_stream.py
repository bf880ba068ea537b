"""Soak run (not part of the test suite): N random STREAMS through parallel.stream_pairs -- per stream random parameters,
6 - 9 pairs whose shape and dtype change once or twice mid-stream (the input slots are re-sized while the pipeline drains),
random pipeline depth, sometimes caller-provided output rows, sometimes an early exit of the consumer -- every result
bit for bit against the oracle orchestration.  python3 tools/soak_stream.py [N] [first seed] [max side]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import parallel, synthetic   # noqa: E402
from oracle import register_oracle as RO           # noqa: E402

n, s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
SMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 1300
bad = 0
t0 = time.time()
for seed in range(s0, s0 + n):
    rng = np.random.default_rng(seed)
    tile = int(rng.integers(100, 600))
    ov = int(rng.integers(10, min(tile // 2 - 1, 100)))
    p = dict(num_pyr_lvl=int(rng.integers(0, 3)), use_full_res_img=bool(rng.integers(0, 2)), use_dog=bool(rng.integers(0, 2)),
             tile_size=tile, overlap=ov, num_iterations=int(rng.integers(1, 4)))
    if p["num_pyr_lvl"] == 0:
        p["use_full_res_img"] = True
    specs = []
    for _ in range(int(rng.integers(1, 4))):          # 1 - 3 runs of equal shape
        H, W = int(rng.integers(420, SMAX)), int(rng.integers(420, SMAX))
        dt = [np.uint8, np.uint16, np.float32][rng.integers(0, 3)]
        specs += [(H, W, dt)] * int(rng.integers(2, 4))
    pairs = [synthetic.make_pair(H, W, seed * 100 + k, dt) for k, (H, W, dt) in enumerate(specs)]
    warp = bool(rng.integers(0, 4))
    depth = int(rng.integers(1, 4))
    stop_after = len(pairs) if rng.integers(0, 5) else int(rng.integers(1, len(pairs)))
    got = []
    gen = parallel.stream_pairs(iter(pairs), p, warp=warp, depth=depth)
    for res in gen:
        got.append((res.flow.copy(), None if res.warped is None else res.warped.copy(), [r.accepted for r in res.reports]))
        if len(got) == stop_after:
            gen.close()
            break
    ok = len(got) == stop_after
    for (flow, warped, acc), (ref, mov) in zip(got, pairs):
        exp, rep = RO.register(ref, mov, nthreads=64, **p)
        ok &= np.array_equal(flow, exp) and acc == [r[3] for r in rep]
        if warp:
            ok &= np.array_equal(warped, RO.warp(mov, exp, tile, ov))
    bad += not ok
    print(seed, [(h, w, np.dtype(d).name) for h, w, d in specs], p, f"depth {depth} warp {warp} consumed {stop_after}/{len(pairs)}",
          "OK" if ok else "MISMATCH", flush=True)
print(f"{n} streams, {bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
