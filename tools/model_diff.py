"""What the choice of rounding model changes (README.md "Tolerances"): register() + warp() of BASELINE cfg2 / cfg3 under the
unfused default and under the FMA models of the window blur (muladd_fused) and the dog() chain (dog_muladd_fused); and how
many gate decisions flip over N random small configurations.  python3 tools/model_diff.py [N soak configs]"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
from microaligner_amd import OptFlowRegistrator, Warper, synthetic   # noqa: E402
from microaligner_amd.device import get_context                      # noqa: E402

ctx = get_context()


def run(ref, mov, params, fb, dg):
    reg = OptFlowRegistrator()
    reg.verbose = False
    for k, v in params.items():
        setattr(reg, k, v)
    reg.muladd_fused, reg.dog_muladd_fused = fb, dg
    reg.ref_img, reg.mov_img = ref, mov
    flow = reg.register()
    w = Warper()
    w.tile_size, w.overlap = reg.tile_size, reg.overlap
    w.image, w.flow = mov, flow
    return flow, w.warp(), [bool(r.accepted) for r in reg.level_reports]


def compare(name, H, params, dtype=np.float32):
    ref, mov = synthetic.make_pair(H, H, 1, dtype)
    base_flow, base_warp, base_acc = run(ref, mov, params, False, False)
    base_flow, base_warp = np.array(base_flow), np.array(base_warp)
    out = {"workload": name, "accepted_default": base_acc}
    for tag, fb, dg in (("muladd_fused", True, False), ("dog_muladd_fused", False, True), ("both", True, True)):
        if dg and not params.get("use_dog") :
            # the gate's dog() images still use the chain: the model matters through the gate only
            pass
        flow, warp, acc = run(ref, mov, params, fb, dg)
        d = np.abs(np.asarray(flow) - base_flow).max(axis=2).ravel()
        dw = np.abs(np.asarray(warp).astype(np.float64) - base_warp.astype(np.float64)).ravel()
        out[tag] = {"flow_delta_px": {"max": float(d.max()), "p99.9": float(np.quantile(d[::7], 0.999)), "mean": float(d.mean())},
                    "warped_delta_grey": {"max": float(dw.max()), "differing_px_share": float((dw != 0).mean())},
                    "gate_decisions_equal": acc == base_acc}
        del flow, warp
    # the dog() image of the full-resolution reference under the two models of the chain
    d0 = ctx.dog_u8(ctx.asdevice(ref), 5, 9, flags=0).numpy()
    d3 = ctx.dog_u8(ctx.asdevice(ref), 5, 9, flags=3).numpy()
    diff = d0.astype(np.int16) - d3.astype(np.int16)
    out["dog_image"] = {"differing_px": int((diff != 0).sum()), "of": int(diff.size), "max_grey_levels": int(np.abs(diff).max())}
    print(json.dumps(out), flush=True)


compare("cfg2 4096^2 f32", 4096, dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=False, tile_size=1000, overlap=100))
compare("cfg3 16384^2 f32 DOG", 16384, dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True, tile_size=1000, overlap=100))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
flips = {"muladd_fused": 0, "dog_muladd_fused": 0, "both": 0}
levels = 0
for seed in range(5000, 5000 + n):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(230, 1500)), int(rng.integers(230, 1500))
    dtype = [np.uint8, np.uint16, np.float32][rng.integers(0, 3)]
    tile = int(rng.integers(100, 700))
    ov = int(rng.integers(10, min(tile // 2 - 1, 110)))
    p = dict(num_pyr_lvl=int(rng.integers(0, 4)), use_full_res_img=bool(rng.integers(0, 2)), use_dog=bool(rng.integers(0, 2)),
             tile_size=tile, overlap=ov, num_iterations=int(rng.integers(1, 4)))
    if p["num_pyr_lvl"] == 0 or min(H, W) / 2 < 100:
        p["use_full_res_img"] = True
    make = synthetic.make_unrelated_pair if rng.integers(0, 4) == 0 else synthetic.make_pair
    ref, mov = make(H, W, seed, dtype)
    _, _, base = run(ref, mov, p, False, False)
    levels += len(base)
    for tag, fb, dg in (("muladd_fused", True, False), ("dog_muladd_fused", False, True), ("both", True, True)):
        _, _, acc = run(ref, mov, p, fb, dg)
        flips[tag] += sum(a != b for a, b in zip(acc, base))
print(json.dumps({"soak_configs": n, "gate_decisions": levels, "flipped_decisions": flips}))
