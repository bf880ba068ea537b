#!/usr/bin/env python3
"""Benchmark of the hot path: OptFlowRegistrator.register() + one Warper.warp(mov, flow).

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4|cfg1]

A step is one registration + warp of one synthetic (ref, mov) pair that is already resident in HBM.
Every rank (one process per GPU) works on its own pair -- independent units, no data-path collective
 -- so scaling is weak and `value` = N * H*W / max-over-ranks time.  The control plane
(barrier, max of the per-rank times) uses torch.distributed/gloo on the host; the GPU is driven only by
libmicroaligner_hip.so.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Mpix/s optical-flow reg+warp, 16k×16k float32 tile, 1/2/4/8 GPU"  # BASELINE.json
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PK_PEAK_TLOPS = 71.4  # packed FP32 lane-ops/s measured with v_pk_mul_f32 at 8 waves/SIMD (profiles/r01_ubench_valu.txt)

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "cfg3": dict(shape=(16384, 16384), params=dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True),
                 desc="16384x16384 f32, DOG preprocess, 5-level pyramid [16,8,4,2,1], tile 1000 / overlap 100 / win 99 / 3 iters"),
    "cfg2": dict(shape=(4096, 4096), params=dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=False),
                 desc="4096x4096 f32, 3 levels [4,2,1], tile 1000 / overlap 100 / win 99 / 3 iters"),
    "cfg4": dict(shape=(8192, 8192), params=dict(num_pyr_lvl=3, use_full_res_img=True, use_dog=False),
                 desc="8192x8192 f32 cycle, shipped YAML parameters, one cycle per GPU"),
    "cfg1": dict(shape=(512, 512), params=dict(), desc="512x512 f32, class defaults (plumbing)"),
}


def algorithmic_bytes_per_px(kernel, iters, esz):
    """Algorithmic HBM bytes per processed pixel and launch (DESIGN.md 'Kernels')."""
    if kernel == "polyexp_m0":      # 2 images in, R0+R1 (2x20 B) and the first M (20 B) out
        return 2 * esz + 60
    if kernel == "blur_v":          # M in; the separable intermediate V is not algorithmic traffic
        return 20
    if kernel == "blur_h_solve":    # non-last: flow (8) + UpdateMatrices (R0 20 + R1 20 + flow 8 + M 20); last: flow 8
        return ((iters - 1) * 76 + 8) / iters
    if kernel == "warp":            # image in/out + flow in
        return 2 * esz + 8
    if kernel == "merge":
        return 24
    if kernel == "pyr_down":
        return 1.25 * esz
    if kernel == "pyr_up":
        return 10
    if kernel == "dog":
        return 17
    if kernel == "nmi":
        return 2
    return 0


def pmc_traffic(workload):
    """HBM bytes per step and kernel group from the committed rocprofv3 PMC summary of this very command
    (profiles/rNN_hbm_traffic_<workload>.json, tools/collect_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the bench."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_{workload}.json")))
    if not files:
        return {}
    return json.load(open(files[-1])).get("per_bench_group_bytes_per_step", {})


def roofline_entry(name, rec, iters, esz, steps=1, traffic_per_step=None, winsize_taps=0):
    if rec["launches"] == 0 or rec["ms"] <= 0:
        return None
    bpp = algorithmic_bytes_per_px(name, iters, esz)
    gbs = bpp * rec["px"] / (rec["ms"] * 1e-3) / 1e9
    traffic = None
    if traffic_per_step and name in traffic_per_step:
        traffic = round(traffic_per_step[name] * steps / rec["launches"])  # HBM bytes per launch (PMC)
    out = {"kernel": name, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
           "avg_launch_ms": round(rec["ms"] / rec["launches"], 4), "launches": rec["launches"],
           "algorithmic_bytes_per_launch": round(bpp * rec["px"] / rec["launches"]),
           "algorithmic_bytes_per_px": round(bpp, 2), "px_per_launch": round(rec["px"] / rec["launches"])}
    if name in ("blur_v", "blur_h_solve") and winsize_taps:
        # informational: these two kernels are bound by packed-FP32 issue, not by HBM.  FIR work = 5 planes x
        # (3 lane-ops per tap pair + 1) per pixel; peak = the v_pk_mul_f32 rate measured on this GPU
        # (profiles/r01_ubench_valu.txt, 8 waves/SIMD)
        lane_ops = 5 * (3 * winsize_taps + 1) * rec["px"]
        tl = lane_ops / (rec["ms"] * 1e-3) / 1e12
        out["valu"] = {"achieved": round(tl, 2), "peak": VALU_PK_PEAK_TLOPS, "unit": "T lane-ops/s (fp32, unfused)",
                       "frac": round(tl / VALU_PK_PEAK_TLOPS, 4)}
    return out


def cpu_baseline(sample, params):
    """The CPU oracle (oracle/, kind 'port') on a bounded sample of the same workload, host cores of this box."""
    import numpy as np  # noqa: F401
    from microaligner_amd import synthetic
    from oracle import register_oracle as RO
    # only the Farneback windows of a level fan out over threads (OpenMP, one window per thread); the largest level
    # of the sample has nwin windows, so that is the number of host threads actually busy
    nwin = (-(-sample // params.get("tile_size", 1000))) ** 2
    cores = min(os.cpu_count() or 1, nwin)
    ref, mov = synthetic.make_pair(sample, sample, 1)
    t0 = time.perf_counter()
    flow, _ = RO.register(ref, mov, nthreads=cores, **params)
    RO.warp(mov, flow, params.get("tile_size", 1000), params.get("overlap", 100))
    dt = time.perf_counter() - t0
    return {"value": round(sample * sample / dt / 1e6, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "sample": f"{sample}x{sample} f32 pair, same parameters as the GPU workload, register()+warp(); the "
                      f"{nwin} Farneback windows of the largest level run on {cores} OpenMP threads (one window per "
                      f"thread, as the reference's dask fan-out; host has {os.cpu_count()} hardware threads), every other "
                      f"stage is single-threaded; {dt:.1f} s of wall time"}


def lanes_leg(lanes, steps, device, dref, dmov, params, tile, overlap):
    """Seconds per pair with `lanes` pairs in flight (each lane: own context, `steps` register()+warp() passes)."""
    import threading
    from microaligner_amd import OptFlowRegistrator, Warper
    from microaligner_amd.device import Context, use_context
    bar, spans = threading.Barrier(lanes), []

    def lane():
        ctx = Context(device)
        with use_context(ctx):
            reg = OptFlowRegistrator()
            reg.verbose = False
            for k, v in params.items():
                setattr(reg, k, v)
            w = Warper()
            w.tile_size, w.overlap = tile, overlap

            def one():
                reg.ref_img, reg.mov_img = dref, dmov   # read-only inputs shared by the lanes
                flow = reg.register()
                w.image, w.flow = dmov, flow
                return w.warp()

            one()
            ctx.sync()
            bar.wait()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            ctx.sync()
            spans.append((t0, time.perf_counter()))
        ctx.close()

    th = [threading.Thread(target=lane) for _ in range(lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if len(spans) != lanes:
        raise RuntimeError("a lane failed")
    return (max(b for _, b in spans) - min(a for a, _ in spans)) / (lanes * steps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=0, help="override H=W of the workload")
    ap.add_argument("--fused", action="store_true", help="window blur with FMA (MA_FB_MULADD_FUSED)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dog", action="store_true", help="experiment: run the workload with use_dog=False")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational FMA-mode leg (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--lanes", type=int, default=3, help="pairs in flight for the informational multi-lane leg (0: skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import numpy as np
    from microaligner_amd import OptFlowRegistrator, Warper, synthetic
    from microaligner_amd.device import get_context

    wl = WORKLOADS[args.workload]
    H, W = (args.size, args.size) if args.size else wl["shape"]
    params = dict(wl["params"])
    if args.no_dog:
        params["use_dog"] = False
    ctx = get_context(local_rank)

    ref, mov = synthetic.make_pair(H, W, seed=1 + rank)
    dref, dmov = ctx.asdevice(ref), ctx.asdevice(mov)
    del ref, mov

    reg = OptFlowRegistrator()
    reg.verbose = False
    reg.muladd_fused = args.fused
    for k, v in params.items():
        setattr(reg, k, v)
    warper = Warper()
    warper.tile_size, warper.overlap = reg.tile_size, reg.overlap

    def step():
        reg.ref_img, reg.mov_img = dref, dmov
        flow = reg.register()
        warper.image, warper.flow = dmov, flow
        return warper.warp()

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_reset()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    ctx.sync()
    t1 = time.perf_counter()
    barrier()
    ctx.profile(False)
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    del out

    if rank == 0:
        prof = ctx.profile_get()
        iters, esz = reg.num_iterations, 4
        # with use_dog the Farneback inputs are the uint8 DOG images (1 B/px), not the f32 level images
        fb_esz = 1 if reg.use_dog else esz
        tps = pmc_traffic(args.workload) if not args.size and not args.fused and not args.no_dog else {}
        win = reg.overlap - (1 - reg.overlap % 2)
        kernels = {k: roofline_entry(k, v, iters, fb_esz if k == "polyexp_m0" else esz, args.steps, tps, win // 2)
                   for k, v in prof.items()}
        kernels = {k: v for k, v in kernels.items() if v}
        total_kernel_ms = sum(v["ms"] for v in prof.values())
        dominant = max(kernels, key=lambda k: prof[k]["ms"]) if kernels else None
        for k in kernels:
            kernels[k]["share_of_kernel_time"] = round(prof[k]["ms"] / total_kernel_ms, 4)
        res = {
            "metric": METRIC, "value": round(world * H * W * args.steps / elapsed / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}" + (f" (size overridden to {H})" if args.size else ""),
                       "pairs_per_step": world, "tile_size": reg.tile_size, "overlap": reg.overlap,
                       "num_iterations": reg.num_iterations, "muladd": "fma" if args.fused else "mul+add",
                       "levels": [[r.factor, r.accepted] for r in reg.level_reports],
                       "parallelism": f"{world} independent pairs, one per GPU, no collective"},
            "roofline": kernels.get(dominant),
            "roofline_polyexp": kernels.get("polyexp_m0"),
            "kernels": kernels,
            "kernel_time_ms_per_step": round(total_kernel_ms / args.steps, 3),
        }
        if world == 1 and not args.fused and not args.no_variants:
            # informational: the same workload with the window blur in the FMA rounding model
            # (MA_FB_MULADD_FUSED: OpenCV builds whose v_muladd is a fused multiply-add); not the headline value
            reg.muladd_fused = True
            step()
            ctx.sync()
            tf0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            ctx.sync()
            tf = (time.perf_counter() - tf0) / args.steps
            reg.muladd_fused = False
            res["variants"] = {"muladd_fma": {"value": round(H * W / tf / 1e6, 2), "unit": "Mpix/s",
                                              "ms_per_step": round(tf * 1e3, 3)}}
            if args.lanes > 1:
                # informational: `lanes` independent pairs in flight on this GPU, one context (HIP stream, workspace)
                # and one host thread per lane -- what parallel.register_pairs(lanes=...) does for a list of pairs
                tl = lanes_leg(args.lanes, args.steps, ctx.device, dref, dmov, params, reg.tile_size, reg.overlap)
                res["variants"][f"lanes{args.lanes}"] = {
                    "value": round(H * W / tl / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(tl * 1e3, 3),
                    "pairs_in_flight": args.lanes}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.cpu_sample, params)
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
