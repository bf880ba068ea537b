#!/usr/bin/env python3
"""Benchmark of the hot path: OptFlowRegistrator.register() + one Warper.warp(mov, flow).

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4|cfg5|cfg1]

A step is one registration + warp of one synthetic (ref, mov) pair that is already resident in HBM.
Every rank (one process per GPU) works on its own pair -- independent units, no data-path collective
 -- so scaling is weak and `value` = N * H*W / max-over-ranks time.  The control plane
(barrier, max of the per-rank times) uses torch.distributed/gloo on the host; the GPU is driven only by
libmicroaligner_hip.so.  Prints ONE JSON line on rank 0.

Launching: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` every process is a
rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Started directly with `--gpus N` (N > 1) and no
WORLD_SIZE in the environment, this process becomes a launcher: it starts N rank processes (the fan-out the
reference gets from dask, shared_modules/utils.py:117-123) BEFORE anything touches HIP, forwards rank 0's JSON
line and exits non-zero if any rank fails.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Mpix/s optical-flow reg+warp, 16k×16k float32 tile, 1/2/4/8 GPU"  # BASELINE.json
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NOMINAL_GHZ = 2.4      # MI355X_MICROARCH.md, chip-level parameters
# FP32 vector peak for the NON-fused arithmetic of the window blurs and the DOG (one flop per lane and instruction
# slot: v_pk_add / v_pk_mul): 256 CUs x 4 SIMDs x 32 lanes per clock = 78.6 TFLOP/s at 2.4 GHz -- half of the 157.3
# TFLOP/s the data sheet quotes for fused multiply-adds.  The chip does not hold 2.4 GHz in these kernels (it is power
# limited: profiles/r03_notes.md); `frac` is taken against the NOMINAL peak; ma_clock_probe measures the clock the chip
# sustains under the same instruction mix in the same run, and the fraction of the peak at THAT clock is reported beside it
# (`frac_at_sustained_clock`).
VALU_LANES_PER_CLOCK = 256 * 4 * 32
VALU_BOUND = ("blur_v", "blur_h_solve", "dog")

CFG2 = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=False)
WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "cfg3": dict(shape=(16384, 16384), params=dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True),
                 desc="16384x16384 f32, DOG preprocess, 5-level pyramid [16,8,4,2,1], tile 1000 / overlap 100 / win 99 / 3 iters"),
    "cfg2": dict(shape=(4096, 4096), params=CFG2,
                 desc="4096x4096 f32, 3 levels [4,2,1], tile 1000 / overlap 100 / win 99 / 3 iters"),
    "cfg4": dict(shape=(8192, 8192), params=dict(num_pyr_lvl=3, use_full_res_img=True, use_dog=False),
                 desc="8192x8192 f32 cycle, shipped YAML parameters (config_1.yaml:42-49), one cycle pair per GPU "
                      "(independent pairs against a fixed reference; the reference's chain is serial)"),
    # BASELINE.json configs[4]: one mosaic tile per step; the affine initialisation is the known synthetic matrix
    # applied with the dense warp kernel (ground-truth matrix where opencv-contrib is absent), then
    # the optical-flow refinement as cfg2
    "cfg5": dict(shape=(4096, 4096), params=CFG2, affine=True,
                 desc="4096x4096 f32 mosaic tile: affine init (known synthetic similarity, warp_affine kernel) then "
                      "optical-flow refinement as cfg2 + warp; tiles dealt round-robin to the GPUs"),
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


# own loads + stores per processed pixel where they differ from the algorithmic bytes (the window blurs: M in + V out;
# V in + R0 + R1 + M out, or V in + flow out in the last iteration)
OWN_IO_BYTES_PER_PX = {"blur_v": 40, "blur_h_solve": lambda iters: ((iters - 1) * 80 + 28) / iters}
FP32_FLOP_PEAK_TF = 157.3


def pmc_traffic(workload):
    """HBM bytes per step and kernel group from the newest committed rocprofv3 PMC summary of this very command
    (profiles/rNN_hbm_traffic_<workload>.json, tools/collect_profiles.sh: separate --pmc FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  Counters cannot be read from inside the bench,
    so the JSON line names the file the bytes come from (`traffic_source`).  A summary is only quoted when it was
    collected with the very kernels that are loaded now: it records the hash of the kernel sources (ma_version()),
    and a summary with another hash -- or none -- yields `traffic: null` and says why."""
    import glob
    from microaligner_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_hbm_traffic_{workload}.json")))
    if not files:
        return {}, None, "no PMC summary committed for this workload"
    rel = os.path.relpath(files[-1], ROOT)
    summary = json.load(open(files[-1]))
    have, want = summary.get("kernel_source_hash"), _lib.source_hash()
    if have != want:
        return {}, None, f"{rel} was collected with kernel sources {have}, the loaded library is {want}"
    return summary.get("per_bench_group_bytes_per_step", {}), rel, None


def kernel_clocks(workload):
    """Effective shader clock per kernel group from the newest committed rocprofv3 summary (tools/kernel_clock.sh:
    GRBM_GUI_ACTIVE / 8 / dispatch duration), quoted only when it was collected with the kernels that are loaded now.
    The blur kernels hold a lower clock than ma_clock_probe's register-only load (they also drive LDS and HBM)."""
    import glob
    from microaligner_amd import _lib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_clocks_{workload}.json")))
    if not files:
        return {}, None
    summary = json.load(open(files[-1]))
    if summary.get("kernel_source_hash") != _lib.source_hash():
        return {}, None
    names = {"fb_blur_h_solve": "blur_h_solve", "fb_blur_v_stream": "blur_v", "fb_blur_v": "blur_v", "dog_fused": "dog"}
    out = {names[k]: v["clock_ghz"] for k, v in summary.get("per_kernel", {}).items() if k in names}
    return out, os.path.relpath(files[-1], ROOT)


def valu_flops_per_px(kernel, winsize_taps, fused, iters=3):
    """FP32 lane-operations per processed pixel of the VALU-bound kernels (what a perfect schedule must still issue).
    Window blurs: 5 planes x (k0 * c, then per tap pair add, multiply, add -- or add, fma).  The horizontal pass also
    carries the rest of A.1 steps 3-4 for the pixel: the 2x2 solve (12 operations, f64) in every iteration and, in all
    but the last, UpdateMatrices (coordinates and weights 8, five bilinear blends 35, the matrix entries 47 = 90).
    DOG: two sigmas x (row filter of 41 taps: 41 multiplies + 40 additions; column filter: 1 multiply + 20 x (add,
    multiply, add)) + the difference + the two normalisations (multiply, add)."""
    if kernel in ("blur_v", "blur_h_solve"):
        fir = 5 * ((2 if fused else 3) * winsize_taps + 1)
        if kernel == "blur_h_solve":
            return fir + 12 + 90 * (iters - 1) / iters
        return fir
    if kernel == "dog":
        return 2 * ((41 + 40) + (1 + 20 * 3)) + 1 + 4
    return 0


def roofline_entry(name, rec, iters, esz, steps=1, traffic_per_step=None, winsize_taps=0, traffic_source=None,
                   clock_ghz=None, fused=False, traffic_note=None, kernel_clock=None, kernel_clock_source=None):
    if rec["launches"] == 0 or rec["ms"] <= 0:
        return None
    bpp = algorithmic_bytes_per_px(name, iters, esz)
    sec = rec["ms"] * 1e-3
    gbs = bpp * rec["px"] / sec / 1e9
    traffic = None
    if traffic_per_step and name in traffic_per_step:
        traffic = round(traffic_per_step[name] * steps / rec["launches"])  # HBM bytes per launch (PMC)
    hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
    own = OWN_IO_BYTES_PER_PX.get(name)
    if callable(own):
        own = own(iters)
    if own:
        # what this kernel's own loads and stores amount to: for the two window blurs that includes the
        # separable intermediate V, which SURVEY 8d does not count as algorithmic but which does cross HBM
        hbm["own_io_bytes_per_px"] = round(own, 2)
        hbm["own_io_gbs"] = round(own * rec["px"] / sec / 1e9, 1)
        hbm["own_io_frac"] = round(own * rec["px"] / sec / 1e9 / HBM_PEAK_GBS, 4)
    out = {"kernel": name, "bound": "hbm", **hbm, "traffic": traffic,
           "traffic_source": traffic_source if traffic is not None else None,
           "avg_launch_ms": round(rec["ms"] / rec["launches"], 4), "launches": rec["launches"],
           "algorithmic_bytes_per_launch": round(bpp * rec["px"] / rec["launches"]),
           "algorithmic_bytes_per_px": round(bpp, 2), "px_per_launch": round(rec["px"] / rec["launches"])}
    if traffic is None and traffic_note:
        out["traffic_note"] = traffic_note
    if traffic is not None and traffic < 0.8 * out["algorithmic_bytes_per_launch"]:
        # fewer HBM bytes than algorithmic bytes: the producer's output is still in L2 / the 256 MB infinity cache and
        # write-backs complete after the kernel; `achieved` (algorithmic bytes per second) can then exceed the HBM peak
        out["note"] = ("PMC traffic below the algorithmic bytes: part of this kernel's data is served by L2 / infinity "
                       "cache; `achieved` is not pure HBM bandwidth")
    if name in VALU_BOUND and (winsize_taps or name == "dog"):
        # these kernels are bound by FP32 vector issue, not by HBM: price them against the VALU peak at the clock the
        # chip sustains under this instruction mix (ma_clock_probe, same run); the HBM view stays in `hbm`
        fpp = valu_flops_per_px(name, winsize_taps, fused, iters)
        tf = fpp * rec["px"] / sec / 1e12
        # headline: against the NOMINAL peak (256 CUs x 4 SIMD-32 at 2.4 GHz, MI355X_MICROARCH.md); beside it the same
        # against the peak at the clock this very kernel held (rocprofv3 summary of the loaded kernels) or, failing that, the
        # clock ma_clock_probe sustained in this run -- the chip is power limited in these kernels (profiles/r03_notes.md)
        ghz = kernel_clock or clock_ghz or NOMINAL_GHZ
        peak_nom = VALU_LANES_PER_CLOCK * NOMINAL_GHZ * 1e9 / 1e12
        peak_clk = VALU_LANES_PER_CLOCK * ghz * 1e9 / 1e12
        out.update({"bound": "valu", "achieved": round(tf, 2), "peak": round(peak_nom, 2), "unit": "TFLOP/s",
                    "frac": round(tf / peak_nom, 4),
                    # the same arithmetic against the guide's FP32 FLOP peak (157.3 TFLOP/s: every lane-operation a fused
                    # multiply-add counted as two) -- the unfused model cannot issue FMAs, so this view caps at 0.5
                    "frac_flops_of_157TF": round(tf / FP32_FLOP_PEAK_TF, 4),
                    "peak_at_sustained_clock": round(peak_clk, 2), "frac_at_sustained_clock": round(tf / peak_clk, 4),
                    "clock_ghz": round(ghz, 3),
                    "clock_source": (f"{kernel_clock_source} (GRBM_GUI_ACTIVE of this kernel)" if kernel_clock else
                                     "ma_clock_probe, same run" if clock_ghz else "nominal"),
                    "clock_probe_ghz": round(clock_ghz, 3) if clock_ghz else None,
                    "flops_per_px": round(fpp, 1), "arithmetic": "fma" if fused else "mul+add (3 ops per tap pair)", "hbm": hbm})
    if name == "warp":
        # instruction bound, not bandwidth bound: VALU busy 0.90 at half the HBM rate (profiles/r05_sq_counters_cfg3.txt), 88
        # VALU instructions per pixel measured (rocprofv3 SQ_INSTS_VALU, profiles/r06_notes.md: coordinate quantisation with
        # cvtss2si semantics, weights from the 5-bit fractions, unfused blend, window-local indices, the by-products).  The
        # bandwidth figures stay as the HBM view; the issue view prices those instructions against one per lane and cycle
        ghz = clock_ghz or NOMINAL_GHZ
        ipp = 88.0
        rate = ipp * rec["px"] / sec / 1e12
        peak_nom = VALU_LANES_PER_CLOCK * NOMINAL_GHZ * 1e9 / 1e12
        out.update({"bound": "valu", "hbm": hbm,
                    "valu": {"instructions_per_px": ipp, "source": "profiles/r06_notes.md (SQ_INSTS_VALU / pixels, cfg2)",
                             "achieved": round(rate, 2), "peak": round(peak_nom, 2), "unit": "T lane-instructions/s",
                             "frac": round(rate / peak_nom, 4), "clock_probe_ghz": round(ghz, 3)}})
    return out


def cpu_ranges(cpus):
    """[0, 1, 2, 8, 9] -> "0-2,8-9" (the sysfs cpulist notation)."""
    out, start, prev = [], None, None
    for c in sorted(cpus):
        if start is None:
            start = prev = c
        elif c == prev + 1:
            prev = c
        else:
            out.append(f"{start}-{prev}" if prev > start else str(start))
            start = prev = c
    if start is not None:
        out.append(f"{start}-{prev}" if prev > start else str(start))
    return ",".join(out)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus():
    """(CPUs this process can actually run on at once, how that was found): the smaller of the scheduler's affinity mask and
    the container's CPU-time quota (cgroup v2 cpu.max / v1 cfs_quota).  The GPU boxes of this pool show 256 hardware threads
    and grant 16 CPUs of time (cpu.max 1600000 100000): 256 worker threads there are 16 cores' worth of work, throttled."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    how = "affinity"
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None and quota < n:
        n, how = max(1, int(quota)), f"cgroup CPU quota ({quota:g} of {os.cpu_count()} hardware threads)"
    return n, how


def cpu_baseline_opencv(sample, params):
    """BASELINE.md section 2, plan A: the reference's orchestration over a LIVE cv2 (oracle/cv2_backend.py: the cv2 calls
    as the reference writes them, Farneback windows and NMI chunks fanned out over one worker per hardware thread the way
    utils.py:117-119 / flow_calc.py:93-97 fan them out over dask processes).  Raises ImportError where no cv2 imports."""
    from microaligner_amd import synthetic
    from oracle import cv2_backend
    cores, cores_how = effective_cpus()
    ref, mov = synthetic.make_pair(sample, sample, 1)
    stages = {}
    t0 = time.perf_counter()
    cv2_backend.register_over_cv2(ref, mov, workers=cores, stage_seconds=stages, **params)
    dt = time.perf_counter() - t0
    info = cv2_backend.build_summary()
    return {"value": round(sample * sample / dt / 1e6, 3), "unit": "Mpix/s", "cores": cores,
            "kind": "port" if info["standin"] else "opencv", "cpu": cpu_model(), "cores_limit": cores_how, "opencv": info,
            "stage_seconds": {k: round(v, 2) for k, v in stages.items()},
            "sample": f"{sample}x{sample} f32 pair, same parameters as the GPU workload, register()+warp(), {dt:.1f} s wall; "
                      f"cv2 {info['version']} called as the reference calls it, {cores} windows / NMI chunks in flight "
                      f"(threads: cv2 releases the GIL), tile cutting / stitching single-threaded numpy as in the reference"}


def cpu_baseline(sample, params):
    """The CPU baseline on a bounded sample of the same workload on every host core of this box.  Plan A when a cv2
    imports: the orchestration over the real OpenCV (kind 'opencv', cpu_baseline_opencv).  Otherwise -- this image and the
    GPU pool have no cv2 -- the CPU oracle (oracle/, kind 'port'): Farneback windows, image rows (DOG, pyramids, remap)
    and NMI chunks fan out over OpenMP threads, one per hardware thread -- the analogue of the reference's dask
    scheduler="processes" fan-out (utils.py:117-119, flow_calc.py:93-97, similarity_scoring.py:44-48)."""
    try:
        return cpu_baseline_opencv(sample, params)
    except ImportError:
        pass
    from microaligner_amd import synthetic
    from oracle import register_oracle as RO
    cores, cores_how = effective_cpus()
    ref, mov = synthetic.make_pair(sample, sample, 1)
    stages = {}
    t0 = time.perf_counter()
    flow, _ = RO.register(ref, mov, nthreads=cores, stage_seconds=stages, **params)
    tw = time.perf_counter()
    RO.warp(mov, flow, params.get("tile_size", 1000), params.get("overlap", 100))
    t1 = time.perf_counter()
    stages["final_warp"] = t1 - tw
    dt = t1 - t0
    nwin = (-(-sample // params.get("tile_size", 1000))) ** 2
    return {"value": round(sample * sample / dt / 1e6, 3), "unit": "Mpix/s", "cores": cores, "kind": "port",
            "cpu": cpu_model(), "cores_limit": cores_how, "stage_seconds": {k: round(v, 2) for k, v in stages.items()},
            "sample": f"{sample}x{sample} f32 pair, same parameters as the GPU workload, register()+warp(), "
                      f"{dt:.1f} s wall; C restatement of the OpenCV / scikit-learn arithmetic (no cv2 in the image) "
                      f"with {cores} OpenMP threads: the Farneback windows of a level ({nwin} at full resolution, one "
                      f"window per thread), image rows of DOG / pyramid / remap and NMI chunks run in parallel; "
                      f"tile cutting / stitching is single-threaded numpy as in the reference"}


def page_warp_leg(n_pages, H, W, tile, overlap):
    """SURVEY 8f-1, warp_and_save_pages (__main__.py:288-302): one device-resident flow applied to `n_pages` uint16 host pages
    (the pipeline's page dtype), pageable numpy arrays in and out, every result page touched before the clock starts.  Two
    callers: Warper.warp_pages (the driver with all pages in hand) and the reference's own loop, one Warper.warp() per
    page.  PCIe inclusive by nature; informational."""
    import numpy as np
    from microaligner_amd import Warper
    from microaligner_amd.device import get_context
    ctx = get_context()
    rng = np.random.default_rng(5)
    base = rng.integers(0, 65535, (H, W), dtype=np.uint16)
    pages = [base ^ np.uint16(257 * k) for k in range(n_pages)]
    out = [np.ones_like(base) for _ in range(n_pages)]
    flow = np.empty((H, W, 2), np.float32)
    flow[..., 0] = 3.3 + np.linspace(-2.0, 2.0, W, dtype=np.float32)
    flow[..., 1] = -2.1 + np.linspace(-1.5, 1.5, H, dtype=np.float32)[:, None]
    dflow = ctx.asdevice(flow)
    del flow
    w = Warper()
    w.tile_size, w.overlap = tile, overlap
    w.flow = dflow
    w.warp_pages(pages[:3], out[:3])        # device slots, staging rings and copy threads exist from here on
    runs = []
    for _ in range(3):
        ctx.transfer_stats(reset=True)
        t0 = time.perf_counter()
        w.warp_pages(pages, out)
        runs.append(time.perf_counter() - t0)
        up, down = ctx.transfer_stats(reset=True)
    t_drv = sorted(runs)[1]                 # the median of three calls

    def loop(k):
        for i in range(k):
            w2 = Warper()
            w2.tile_size, w2.overlap = tile, overlap
            w2.image, w2.flow = pages[i], dflow
            res = w2.warp()
        return res
    loop(min(3, n_pages))                   # the pool's page-locked result buffers exist from here on
    t0 = time.perf_counter()
    last = loop(n_pages)
    t_loop = time.perf_counter() - t0
    same = bool(np.array_equal(last, out[n_pages - 1]))
    px = n_pages * H * W
    return {"value": round(px / t_drv / 1e6, 1), "unit": "Mpix/s", "pages": n_pages, "page": f"{H}x{W} uint16",
            "ms_per_page": round(t_drv / n_pages * 1e3, 2), "ms_per_call": [round(t * 1e3, 1) for t in runs],
            "pcie_gb_s_both_directions": round((up + down) / t_drv / 1e9, 1),
            "per_page_warp_loop": {"value": round(px / t_loop / 1e6, 1), "unit": "Mpix/s",
                                   "ms_per_page": round(t_loop / n_pages * 1e3, 2), "same_pixels_as_driver": same},
            "what": "Warper.warp_pages: one resident flow, pageable numpy pages in, pageable numpy pages out (upload, kernel and "
                    "download overlapped in bands of tile rows on the three engines); per_page_warp_loop: the reference's loop, "
                    "one Warper.warp() per page (the same driver, one page per call, page-locked result from the pool)"}


def lanes_leg(lanes, steps, device, dref, dmov, params, tile, overlap):
    """Seconds per pair with `lanes` pairs in flight (each lane: own context, `steps` register()+warp() passes)."""
    import threading
    from microaligner_amd import OptFlowRegistrator, Warper
    from microaligner_amd.device import Context, use_context
    bar, spans, errors = threading.Barrier(lanes), [], []

    def lane():
        ctx = None
        try:
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
                bar.wait(timeout=600)
                t0 = time.perf_counter()
                for _ in range(steps):
                    one()
                ctx.sync()
                spans.append((t0, time.perf_counter()))
        except BaseException as e:  # a failed lane must not leave the others waiting at the barrier
            errors.append(e)
            bar.abort()
        finally:
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=lane) for _ in range(lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors or len(spans) != lanes:
        first = next((e for e in errors if not isinstance(e, threading.BrokenBarrierError)), errors[0] if errors else None)
        raise RuntimeError(f"a lane failed: {first!r}")
    return (max(b for _, b in spans) - min(a for a, _ in spans)) / (lanes * steps)


def tile_lanes_leg(lanes, steps, device, work, params, use_features, replay_rounds=True):
    """Seconds per mosaic tile (cfg5) with `lanes` tiles in flight on one GPU: each lane -- own context (HIP stream, workspace,
    pools), own host thread -- takes every lanes-th tile of `work` and runs the whole composition on it (affine initialisation,
    transform, optical-flow refinement, warp), `steps` passes over its share.  A 4096^2 tile leaves most of the chip idle at
    its coarse levels and during the feature stage's synchronisations; a second tile fills that."""
    import threading
    import numpy as np
    from microaligner_amd import FeatureRegistrator, OptFlowRegistrator, Warper
    from microaligner_amd.device import Context, use_context
    lanes = max(1, min(lanes, len(work)))
    bar, spans, errors = threading.Barrier(lanes), [], []

    def lane(i):
        ctx = None
        try:
            ctx = Context(device)
            with use_context(ctx):
                items = work[i::lanes]
                freg = None
                if use_features:
                    freg = FeatureRegistrator()
                    freg.verbose = False
                    freg.skip_repeated_rounds = replay_rounds
                reg = OptFlowRegistrator()
                reg.verbose = False
                for k, v in params.items():
                    setattr(reg, k, v)
                w = Warper()
                w.tile_size, w.overlap = reg.tile_size, reg.overlap

                def one(dref, dmov, inv_affine, _host):
                    if freg is not None:
                        freg.ref_img, freg.mov_img = dref, dmov
                        t_mat = freg.register()
                        m = ctx.warp_affine(dmov, np.linalg.pinv(np.vstack([t_mat, [0, 0, 1]])))
                    else:
                        m = ctx.warp_affine(dmov, inv_affine)
                    reg.ref_img, reg.mov_img = dref, m
                    flow = reg.register()
                    w.image, w.flow = m, flow
                    return w.warp()

                for it in items:
                    one(*it)
                ctx.sync()
                bar.wait(timeout=600)
                t0 = time.perf_counter()
                for _ in range(steps):
                    for it in items:
                        one(*it)
                ctx.sync()
                spans.append((t0, time.perf_counter()))
        except BaseException as e:  # a failed lane must not leave the others waiting at the barrier
            errors.append(e)
            bar.abort()
        finally:
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=lane, args=(i,)) for i in range(lanes)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors or len(spans) != lanes:
        first = next((e for e in errors if not isinstance(e, threading.BrokenBarrierError)), errors[0] if errors else None)
        raise RuntimeError(f"a lane failed: {first!r}")
    return (max(b for _, b in spans) - min(a for a, _ in spans)) / (len(work) * steps), lanes


def host_inclusive_leg(steps, ref, mov, params):
    """numpy in -> numpy out through the drop-in API (optflow_registrator.py:93,173; warper.py:53): H2D of both
    images, register(), warp(), D2H of the flow and the warped image, per pair."""
    from microaligner_amd import OptFlowRegistrator, Warper
    reg = OptFlowRegistrator()
    reg.verbose = False
    for k, v in params.items():
        setattr(reg, k, v)
    w = Warper()
    w.tile_size, w.overlap = reg.tile_size, reg.overlap

    from microaligner_amd.device import get_context
    ctx = get_context()

    def one():
        # a pipeline registers a NEW pair every time: nothing of the previous pass may count as already uploaded.  Within
        # the pass the library does recognise the arrays it has just moved (mov, and the flow register() returned)
        ctx.forget_host_arrays()
        reg.ref_img, reg.mov_img = ref, mov
        flow = reg.register()
        w.image, w.flow = mov, flow
        return flow, w.warp()

    # two untimed passes bound to the same names as the timed loop: the page-locked result pool reaches its steady
    # state (two flow-sized buffers alternate, the previous result is still referenced while the next one is made)
    flow, warped = one()
    flow, warped = one()
    t0 = time.perf_counter()
    for _ in range(steps):
        flow, warped = one()
    dt = (time.perf_counter() - t0) / steps
    assert flow.shape == ref.shape + (2,) and warped.shape == ref.shape
    return dt


def host_stream_leg(n_pairs, ref, mov, params, dtype):
    """A STREAM of distinct numpy pairs through parallel.stream_pairs (numpy in -> numpy out, the way real inputs arrive:
    __main__.py:398-433 reads pages from TIFF): one context, H2D of pair k+1 and D2H of pair k-1 under the kernels of pair
    k.  Returns ms per pair sustained and the per-engine busy times per pair.  The pairs are made distinct by rolling the
    synthetic pair along y (same statistics, different bytes: nothing can be recognised as already uploaded)."""
    import numpy as np
    from microaligner_amd import parallel
    from microaligner_amd.device import get_context
    ctx = get_context()
    ctx.forget_host_arrays()

    def cast(a):
        return a if a.dtype == dtype else np.clip(np.rint(a), 0, 255).astype(dtype)

    pairs = [(cast(np.roll(ref, 53 * k, axis=0)), cast(np.roll(mov, 53 * k, axis=0))) for k in range(n_pairs)]
    for _ in parallel.stream_pairs(pairs[:6], params, warp=True):      # warm-up: slots, page-locked result buffers, both lanes' pools
        pass
    stats = {}
    t0 = time.perf_counter()
    checks, marks = [], []
    for res in parallel.stream_pairs(pairs, params, warp=True, stats=stats):
        marks.append(time.perf_counter())
        checks.append((float(res.flow[::997, ::991].sum()), int(res.warped[::997, ::991].astype(np.int64).sum())))
    wall = (marks[-1] - t0) / n_pairs               # includes filling and draining the pipeline once
    # sustained: the arrival interval over the later half of the stream.  With L compute lanes results arrive in groups of L,
    # so intervals are taken L results apart; the median keeps one hiccup of the host out of the figure
    half, L_ = n_pairs // 2, max(1, int(stats.get("compute_lanes", 1)))
    spans = sorted((marks[i] - marks[i - L_]) / L_ for i in range(max(half, L_), n_pairs))
    dt = spans[len(spans) // 2] if len(spans) % 2 else 0.5 * (spans[len(spans) // 2 - 1] + spans[len(spans) // 2])
    in_bytes = sum(a.nbytes + b.nbytes for a, b in pairs)
    H, W = ref.shape
    out_bytes = n_pairs * (H * W * 8 + H * W * np.dtype(dtype).itemsize)
    tl = stats["timeline"]
    return {"value": round(H * W / dt / 1e6, 2), "unit": "Mpix/s", "ms_per_pair": round(dt * 1e3, 3),
            "arrival_ms": [round((m - t0) * 1e3, 1) for m in marks],
            "h2d_ms": [round((t["h2d"][1] - t["h2d"][0]) * 1e3, 1) for t in tl],
            "ms_per_pair_incl_fill_and_drain": round(wall * 1e3, 3), "pairs": n_pairs,
            "dtype": np.dtype(dtype).name,
            "engine_busy_ms_per_pair": {k: round(stats[k] / n_pairs, 3) for k in ("h2d_busy_ms", "compute_busy_ms", "d2h_busy_ms")},
            "compute_lanes": stats["compute_lanes"],
            "engine_note": "compute_busy_ms is the GPU time from the first to the last kernel of a pair; with two compute lanes two "
                           "pairs are in flight at once (the coarse levels of one under the full-resolution level of the "
                           "other), so it exceeds the arrival interval",
            "every_byte_moved_once": stats["h2d_bytes"] == in_bytes and stats["d2h_bytes"] == out_bytes,
            "h2d_gb_per_pair": round(stats["h2d_bytes"] / n_pairs / 1e9, 3), "d2h_gb_per_pair": round(stats["d2h_bytes"] / n_pairs / 1e9, 3),
            "distinct_results": len(set(checks)),
            "what": "parallel.stream_pairs over distinct numpy pairs: upload / kernels / download of consecutive pairs "
                    "overlapped on the transfer engines of one context and its compute lanes, results delivered as numpy "
                    "arrays in input order"}


HOST_MODES = ("stream_pairs_pageable", "stream_pairs_page_locked", "warp_pages_pageable", "warp_pages_page_locked")


def available_host_bytes():
    """(bytes of host memory this process tree can still take, how that was found): the smaller of the kernel's MemAvailable
    and the container's memory limit minus what it already uses (cgroup v2 memory.max / v1 limit_in_bytes).  A rank that
    runs the node out of memory is killed without a word and takes the whole job -- the headline with it."""
    avail, how = None, "unknown"
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail, how = int(line.split()[1]) * 1024, "MemAvailable"
                break
    except OSError:
        pass
    for lim_f, cur_f in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                         ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            lim = open(lim_f).read().strip()
            if lim == "max":
                break
            room = int(lim) - int(open(cur_f).read())
            if room < 2 ** 60 and (avail is None or room < avail):
                avail, how = max(0, room), "cgroup memory limit"
            break
        except (OSError, ValueError):
            continue
    return avail, how


def host_modes_bytes(H, W, itemsize, n_pages):
    """Host memory one rank's host_modes_measure holds at its peak: the two input pairs and two output sets of the stream
    (or the pages in and out + the caller's pair), the library's staging rings, and slack."""
    px = H * W
    stream = 4 * px * itemsize + 2 * (px * 8 + px * itemsize)
    paged = 2 * px * itemsize + (2 * n_pages + 1) * px * 2
    return max(stream, paged) + (3 << 30)


class SoftSync:
    """Barriers and the exchange of result rows for the host modes over a key-value store of their own (a TCPStore that rank 0
    hosts), every wait bounded.  The headline has been gathered before these modes start; whatever happens in them -- a rank
    that fails half way, one that is slow beyond reason -- costs at most `timeout` seconds and a row that says so, never the
    line: a torch.distributed collective entered by some ranks only waits for half an hour and takes the job with it
    (a 4-rank rehearsal did exactly that, profiles/r06_notes.md)."""

    def __init__(self, store, world, rank, timeout):
        self.store, self.world, self.rank, self.timeout, self.n, self.name = store, world, rank, float(timeout), 0, "main"

    def section(self, name):
        """Barriers are numbered within a section: a rank that left one section early meets the others again in the next."""
        self.name, self.n = name, 0
        return self

    def barrier(self):
        self.n += 1
        key = f"ma/{self.name}/barrier/{self.n}"
        self.store.add(key, 1)
        deadline = time.monotonic() + self.timeout
        while int(self.store.add(key, 0)) < self.world:
            if time.monotonic() > deadline:
                raise TimeoutError(f"{self.name}: barrier {self.n} not reached by every rank within {self.timeout:.0f} s")
            time.sleep(0.002)

    def put_rows(self, rows):
        self.store.set(f"ma_hm/rows/{self.rank}", json.dumps(rows))

    def get_rows(self):
        """rank 0: every rank's rows, an error row for a rank that did not deliver in time"""
        from datetime import timedelta
        out = []
        deadline = time.monotonic() + self.timeout
        for r in range(self.world):
            key = f"ma_hm/rows/{r}"
            try:
                self.store.wait([key], timedelta(seconds=max(1.0, deadline - time.monotonic())))
                out.append(json.loads(self.store.get(key)))
            except Exception as e:   # noqa: BLE001
                out.append({"host_modes_error": f"rank {r} delivered no rows: {e!r}"})
        return out


def soft_sync(dist, world, rank, timeout, port):
    """The key-value store of the informational legs (rank 0 hosts it on `port`, agreed on by a broadcast while the ranks
    were still in step).  None at N = 1."""
    if dist is None:
        return None
    from datetime import timedelta
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(port), world, rank == 0,
                          timeout=timedelta(seconds=timeout), wait_for_workers=False)
    return SoftSync(store, world, rank, timeout)


def host_modes_exchange(soft, world, rank, run):
    """`run(sync, rows)` measures this rank's host modes into `rows` (a barrier = `sync()`); every rank's rows come back on
    rank 0 (None on the others).  No torch.distributed collective in here: barriers and rows go through SoftSync, so a rank
    that fails or stalls costs the others one timeout at most and shows up as an error row."""
    mine = {}
    try:
        if world > 1 and soft is None:
            raise RuntimeError("no store for the informational legs")
        run(soft.section("host modes").barrier if soft is not None else (lambda: None), mine)   # fills `mine` as it goes
    except Exception as e:   # noqa: BLE001 -- a barrier the others did not reach in time, the store itself: this rank leaves the modes
        import traceback
        mine["host_modes_error"] = repr(e) + " | " + traceback.format_exc().strip().splitlines()[-1]
    if soft is None:
        return [mine] if world == 1 else None
    try:
        soft.put_rows(mine)
        return soft.get_rows() if rank == 0 else None
    except Exception as e:   # noqa: BLE001
        return [{"host_modes_error": f"row exchange failed: {e!r}"}] * world if rank == 0 else None


def host_modes_measure(ctx, ref, mov, params, n_pairs, n_pages, sync, res=None):
    """The numpy -> numpy modes of DESIGN.md section 6 on THIS rank, each between two barriers (`sync`) so that every rank of
    the node runs the same mode at the same time -- these, not the device-resident headline, are what can fail to scale: the
    staged path makes three to four passes over host DRAM per payload byte.  Per mode: units, seconds (fill and drain of the
    pipeline included), payload bytes in + out, and whether the library moved the caller's buffers by DMA as they are
    (`direct_*`, the library's own decision: ma_host_transfer_is_direct).
      stream_pairs_*: parallel.stream_pairs over `n_pairs` pairs (two distinct pairs alternating), flow + warped image into
                      caller arrays; *_pageable: plain numpy arrays in and out; *_page_locked: the same arrays page-locked in
                      place (device.host_register; its cost is reported as host_register_ms_per_gib);
      warp_pages_*  : Warper.warp_pages, one resident flow over `n_pages` uint16 pages, pageable / page-locked.
    Every statement that can fail runs inside `guarded`: a failure becomes the row of the mode (or of its set-up) and this rank
    still passes every barrier the others wait at."""
    import numpy as np
    from microaligner_amd import Warper, device, parallel
    H, W = ref.shape
    res, box = ({} if res is None else res), {}

    def guarded(tag, fn, needs=()):
        for k in needs:
            if k not in box:
                res.setdefault(tag, {}).setdefault("error", f"not run: the set-up of {k!r} failed")
                return None
        try:
            return fn()
        except Exception as e:   # noqa: BLE001
            res.setdefault(tag, {})["error"] = repr(e)
            return None

    def setup_stream():
        ins = [(ref, mov), (np.roll(ref, 53, axis=0), np.roll(mov, 53, axis=0))]
        outs = [(np.zeros((H, W, 2), np.float32), np.zeros((H, W), ref.dtype)) for _ in range(2)]   # touched: no page faults timed
        box["ins"], box["outs"], box["seq"] = ins, outs, [ins[k % 2] for k in range(n_pairs)]

    def stream(tag):
        def warm():
            outs = box["outs"]
            for _ in parallel.stream_pairs(box["seq"][:2], params, warp=True, out=lambda i: outs[i % 2]):   # slots, rings, copy threads
                pass

        def timed():
            ins, outs = box["ins"], box["outs"]
            ctx.transfer_stats(reset=True)
            c0, t0 = time.process_time(), time.perf_counter()
            for _ in parallel.stream_pairs(box["seq"], params, warp=True, out=lambda i: outs[i % 2]):
                pass
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            up, down = ctx.transfer_stats(reset=True)
            res[tag] = {"units": n_pairs, "unit": "pair", "seconds": dt, "payload_bytes": int(up + down), "host_cpu_seconds": cpu,
                        "mode_in": device.transfer_mode(ins[0][0]), "mode_out": device.transfer_mode(outs[0][0]),
                        "direct_in": device.transfer_is_direct(ins[0][0]), "direct_out": device.transfer_is_direct(outs[0][0])}
        guarded(tag, warm, needs=("seq",))
        sync()
        if "error" not in res.get(tag, {}):
            guarded(tag, timed, needs=("seq",))
        sync()

    guarded("setup_stream_pairs", setup_stream)
    stream("stream_pairs_pageable")
    guarded("transient_page_locking", lambda: res.__setitem__("transient_page_locking", device.transient_pin_stats()))

    def lock_stream():
        t0 = time.perf_counter()
        locked = [a for pair in box["ins"] + box["outs"] for a in pair]
        box["locked"] = locked
        box["locked_ok"] = all([device.host_register(a) for a in locked])
        res["host_register_ms_per_gib"] = round((time.perf_counter() - t0) * 1e3 / (sum(a.nbytes for a in locked) / 2 ** 30), 1)
    guarded("stream_pairs_page_locked", lock_stream, needs=("seq",))
    stream("stream_pairs_page_locked")
    res.setdefault("stream_pairs_page_locked", {})["registered_in_place"] = bool(box.get("locked_ok"))

    def setup_pages():
        for a in box.pop("locked", []):
            device.host_unregister(a)
        flow_host = box["outs"][0][0]
        for k in ("ins", "seq"):
            box.pop(k, None)
        rng = np.random.default_rng(5)
        base = rng.integers(0, 65535, (H, W), dtype=np.uint16)
        pages = [base ^ np.uint16(257 * k) for k in range(n_pages)]
        pout = [np.ones_like(base) for _ in range(n_pages)]
        w = Warper()
        w.tile_size, w.overlap = params.get("tile_size", 1000), params.get("overlap", 100)
        w.flow = ctx.asdevice(flow_host)          # the flow of the last streamed pair
        box.pop("outs", None)
        box["pages"], box["pout"], box["warper"] = pages, pout, w
    guarded("setup_warp_pages", setup_pages, needs=("outs",))

    def paged(tag):
        def timed():
            pages, pout, w = box["pages"], box["pout"], box["warper"]
            ctx.transfer_stats(reset=True)
            c0, t0 = time.process_time(), time.perf_counter()
            w.warp_pages(pages, pout)
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            up, down = ctx.transfer_stats(reset=True)
            res[tag] = {"units": n_pages, "unit": "uint16 page", "seconds": dt, "payload_bytes": int(up + down), "host_cpu_seconds": cpu,
                        "mode_in": "direct" if device.transfer_is_direct(pages[0]) else "staged (pieces: no transient page-locking)",
                        "mode_out": "direct" if device.transfer_is_direct(pout[0]) else "staged (pieces: no transient page-locking)",
                        "direct_in": device.transfer_is_direct(pages[0]), "direct_out": device.transfer_is_direct(pout[0])}
        guarded(tag, lambda: box["warper"].warp_pages(box["pages"][:2], box["pout"][:2]), needs=("warper",))
        sync()
        if "error" not in res.get(tag, {}):
            guarded(tag, timed, needs=("warper",))
        sync()

    paged("warp_pages_pageable")
    guarded("warp_pages_page_locked",
            lambda: box.__setitem__("pages_ok", all([device.host_register(a) for a in box["pages"] + box["pout"]])), needs=("warper",))
    paged("warp_pages_page_locked")
    res.setdefault("warp_pages_page_locked", {})["registered_in_place"] = bool(box.get("pages_ok"))
    for a in box.get("pages", []) + box.get("pout", []):
        guarded("cleanup", lambda a=a: device.host_unregister(a))
    return res


def host_modes_report(rows, H, W):
    """Rank 0: every rank's host-mode rows -> per mode the per-rank times, the whole-node rate (all units / the slowest rank's
    time) and the whole-node payload rate over PCIe (the staged path moves 3 - 4 times that through host DRAM)."""
    out = {}
    for mode in HOST_MODES:
        per = [r.get(mode) for r in rows]
        good = [p for p in per if p is not None and "seconds" in p]
        if len(good) < len(per):
            # a rank without a measurement: its reason (the mode's own failure, the failed set-up, or why the rank left the modes)
            out[mode] = {"error": [None if (p is not None and "seconds" in p) else
                                   ((p or {}).get("error") or r.get("host_modes_error") or "no row") for p, r in zip(per, rows)]}
            if not good:
                continue
            out[mode]["ranks_measured"] = [i for i, p in enumerate(per) if p is not None and "seconds" in p]
        slowest = max(p["seconds"] for p in good)
        units = sum(p["units"] for p in good)
        out.setdefault(mode, {}).update({
                     "unit": good[0]["unit"], "units_per_rank": good[0]["units"],
                     "ms_per_unit_per_rank": [round(p["seconds"] / p["units"] * 1e3, 2) for p in good],
                     "aggregate_mpix_s": round(units * H * W / slowest / 1e6, 1),
                     "aggregate_payload_gb_s": round(sum(p["payload_bytes"] for p in good) / slowest / 1e9, 1),
                     "buffers_moved_directly": [bool(p["direct_in"] and p["direct_out"]) for p in good],
                     "transfer_mode_in_out": [f"{p.get('mode_in')} / {p.get('mode_out')}" for p in good],
                     "host_cpu_s_per_unit_per_rank": [round(p.get("host_cpu_seconds", 0.0) / p["units"], 4) for p in good]})
        if "registered_in_place" in good[0]:
            out[mode]["registered_in_place"] = [bool(p.get("registered_in_place")) for p in good]
    for extra in ("setup_stream_pairs", "setup_warp_pages", "cleanup", "host_modes_error"):
        errs = {i: (r[extra].get("error") if isinstance(r[extra], dict) else r[extra]) for i, r in enumerate(rows) if r.get(extra)}
        if errs:
            out.setdefault("errors", {})[extra] = {f"rank {i}": e for i, e in errs.items()}
    tp = [r.get("transient_page_locking") for r in rows if r.get("transient_page_locking") is not None
          and "error" not in r["transient_page_locking"]]
    if tp:
        out["transient_page_locking_per_rank"] = tp
    reg = [r.get("host_register_ms_per_gib") for r in rows if r.get("host_register_ms_per_gib") is not None]
    if reg:
        out["host_register_ms_per_gib"] = reg
    out["what"] = ("every rank runs the same numpy -> numpy mode at the same time (barriers on both sides); seconds include "
                   "filling and draining the pipeline once; aggregate_* = all ranks' units (bytes) / the slowest rank's time; "
                   "DESIGN.md section 6 gives the host DRAM passes behind each mode")
    return out


# ---- launcher ----------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """Start n rank processes of this script and forward rank 0's stdout.  Nothing in this process has touched HIP
    (no microaligner_amd import, no torch.cuda call) -- the ranks are plain children, never an exec of a process
    that initialised the GPU.  Rank 0's stdout is drained by a reader thread while the ranks run, so a chatty rank
    can never fill the pipe and block."""
    import threading
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with status {code}; stopping the other ranks", file=sys.stderr)
                for q in pending:
                    procs[q].terminate()   # exact children of this process, by handle
        if pending:
            time.sleep(0.05)
    reader.join(timeout=30)
    out = b"".join(chunks).decode()
    json_lines = [line for line in out.splitlines() if line.startswith("{")]
    for line in out.splitlines():      # anything else rank 0 wrote to stdout (library chatter) goes to stderr
        if not line.startswith("{"):
            print(line, file=sys.stderr)
    sys.stdout.write("".join(line + "\n" for line in json_lines))
    sys.stdout.flush()
    if rc == 0 and not json_lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def finish(dist, rank):
    """The line is out: a rank that does not come to this barrier within a minute is left behind, not waited for."""
    if dist is None:
        return
    try:
        from datetime import timedelta
        dist.monitored_barrier(timeout=timedelta(seconds=60))
        dist.destroy_process_group()
    except Exception as e:   # noqa: BLE001
        print(f"bench.py: rank {rank}: final barrier: {e!r}", file=sys.stderr)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def rank_table(dist, world, rank, row):
    """Every rank's row (a small dict) on rank 0, in rank order: per-rank times, device identity, HBM state."""
    if dist is None:
        return [row]
    rows = [None] * world
    dist.all_gather_object(rows, row)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=0, help="override H=W of the workload")
    ap.add_argument("--dtype", default="f32", choices=["f32", "u8"], help="input dtype (u8: the pipeline-faithful cfg4 variant)")
    ap.add_argument("--fused", action="store_true", help="window blur with FMA (MA_FB_MULADD_FUSED)")
    ap.add_argument("--dog-fused", action="store_true", help="dog() chain with FMA (MA_DOG_FUSED_BLUR | MA_DOG_FUSED_SCALE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dog", action="store_true", help="experiment: run the workload with use_dog=False")
    ap.add_argument("--no-variants", action="store_true", help="skip the informational legs (profiling runs)")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="edge of the CPU-baseline sample; 0: the full workload on hosts with >= 128 hardware threads "
                         "(the 16384^2 oracle run takes ~45 s there), else 4096")
    ap.add_argument("--lanes", type=int, default=3, help="pairs in flight for the informational multi-lane leg (0: skip)")
    ap.add_argument("--stream-pairs", type=int, default=12,
                    help="distinct pairs of the informational numpy -> numpy stream leg (parallel.stream_pairs; 0: skip)")
    ap.add_argument("--page-warps", type=int, default=8,
                    help="uint16 pages of the informational page-warp leg (SURVEY 8f-1: Warper.warp_pages and the per-page "
                         "Warper.warp() loop, numpy in -> numpy out; 0: skip)")
    ap.add_argument("--no-shared-results", action="store_true",
                    help="N > 1: skip writing every rank's last flow / warped image into the node-wide shared array")
    ap.add_argument("--no-companion", action="store_true",
                    help="MA_OPT_COMPANION_STREAM = 0: every kernel alone on the chip (per-kernel timings comparable from "
                         "run to run; the step is ~3 ms longer)")
    ap.add_argument("--feature-init", action="store_true",
                    help="cfg5: FeatureRegistrator.register() supplies the affine initialisation inside the timed step")
    ap.add_argument("--feature-init-from-host", action="store_true",
                    help="with --feature-init: hand FeatureRegistrator the numpy arrays (uploaded inside the timed step) instead "
                         "of the device-resident pair")
    ap.add_argument("--pairs-total", type=int, default=0,
                    help="a step is one pass over K independent pairs dealt round-robin to the ranks (BASELINE cfg4: 8 "
                         "cycle pairs, cfg5: 64 mosaic tiles) instead of one pair per rank; scaling is then strong")
    ap.add_argument("--host-modes", default="auto", choices=["auto", "on", "off"],
                    help="after the timed loop every rank runs the numpy -> numpy modes (parallel.stream_pairs and "
                         "Warper.warp_pages, pageable and page-locked buffers) at the same time and the line reports them per "
                         "rank with the whole-node rates (`host_modes`); auto: when N > 1 -- the modes that can fail to "
                         "scale -- at N = 1 the `variants` legs cover them")
    ap.add_argument("--host-mode-pairs", type=int, default=8, help="pairs per rank of the stream_pairs host modes")
    ap.add_argument("--recompute-rounds", action="store_true",
                    help="cfg5 --feature-init: FeatureRegistrator recomputes a round that follows a rejected one (default: replayed)")
    ap.add_argument("--host-mode-pages", type=int, default=8, help="uint16 pages per rank of the warp_pages host modes")
    ap.add_argument("--host-mode-timeout", type=float, default=300.0,
                    help="seconds a rank waits for the others at a barrier of the host modes before it leaves them (the headline "
                         "has been gathered by then)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / reduce plumbing only, no GPU work (CPU test of the N-rank launcher)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus={world}", file=sys.stderr)
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # gloo announces its connections on the C stdout: keep stdout for the one JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            from datetime import timedelta
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(minutes=10))
            dist.barrier()
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    wl = WORKLOADS[args.workload]
    H, W = (args.size, args.size) if args.size else wl["shape"]
    params = dict(wl["params"])
    if args.no_dog:
        params["use_dog"] = False
    # the pairs of this rank: its own pair (weak scaling), or its share of K pairs dealt round-robin (strong scaling)
    my_pairs = list(range(rank, args.pairs_total, world)) if args.pairs_total else [rank]
    pairs_per_step = args.pairs_total or world

    def reduce_max(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if args.dry_run:
        if dist is not None:
            dist.barrier()
        mine = 1e-3 * args.steps * (1 + 0.01 * rank)
        elapsed = reduce_max(mine)
        # the data plane of a sharded run without GPUs: units are LOADERS (evaluated in the owning rank only), results are
        # written by each rank into a node-wide shared array (parallel.shared_array: POSIX shared memory), nothing but a
        # barrier crosses the control plane; rank 0 then finds every row written by the rank that owns it
        import numpy as np
        from microaligner_amd import parallel
        n_units = pairs_per_step
        loaded = []
        name = f"ma_bench_dry_{os.environ.get('MASTER_PORT', os.getpid())}"
        store = parallel.shared_array(name, (n_units, 4, 8), np.float32, unlink=True)

        def loader(i):
            def load():
                loaded.append(i)
                return i
            return load

        parallel.run_sharded([loader(i) for i in range(n_units)], lambda u: np.full((4, 8), 1000 * rank + u, np.float32),
                             out=store)
        shared_ok = bool(all(np.all(store[i] == 1000 * (i % world) + i) for i in range(n_units)))
        row = {"rank": rank, "ms_per_step": mine / args.steps * 1e3, "pairs": my_pairs,
               "device": None, "pci_bus_id": None, "units_loaded": sorted(loaded)}
        want_modes = args.host_modes == "on" or (args.host_modes == "auto" and world > 1)
        rows = rank_table(dist, world, rank, row)
        host_rows_all = None
        if want_modes:       # the plumbing of the host-mode rows (soft barriers, row exchange, report) with made-up times
            port = [free_port() if (dist is not None and rank == 0) else 0]
            if dist is not None:
                dist.broadcast_object_list(port, src=0)

            def fake_modes(sync, hr):
                # MA_BENCH_DRY_FAIL="<rank>:<mode index>": that rank fails before that mode's barrier (tests: the line survives)
                fail = os.environ.get("MA_BENCH_DRY_FAIL", "")
                fr, fk = (int(v) for v in fail.split(":")) if fail else (-1, -1)
                for k, mode in enumerate(HOST_MODES):
                    if rank == fr and k == fk:
                        raise RuntimeError("injected failure of the dry run")
                    sync()
                    n = args.host_mode_pairs if mode.startswith("stream") else args.host_mode_pages
                    hr[mode] = {"units": n, "unit": "pair" if mode.startswith("stream") else "uint16 page",
                                "seconds": 1e-3 * n * (k + 1) * (1 + 0.01 * rank), "payload_bytes": 1000 * n, "host_cpu_seconds": 1e-4 * n,
                                "mode_in": "direct" if mode.endswith("locked") else "staged", "mode_out": "direct" if mode.endswith("locked") else "staged",
                                "direct_in": mode.endswith("locked"), "direct_out": mode.endswith("locked")}
                hr["host_register_ms_per_gib"] = 100.0
            try:
                soft = soft_sync(dist, world, rank, args.host_mode_timeout, port[0])
            except Exception as e:   # noqa: BLE001
                soft = None
                print(f"bench.py: rank {rank}: store of the informational legs: {e!r}", file=sys.stderr)
            host_rows_all = host_modes_exchange(soft, world, rank, fake_modes)
        rows = rank_table(dist, world, rank, row)
        if rank == 0:
            per = [r["ms_per_step"] for r in rows]
            print(json.dumps({"metric": METRIC, "value": None, "unit": "Mpix/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "dry_run": True,
                              "config": {"workload": args.workload, "pairs_per_step": pairs_per_step},
                              "ranks": rows, "results_via": "shared memory array written in place by the owning rank",
                              "shared_results_ok": shared_ok,
                              "host_modes": host_modes_report(host_rows_all, 1000, 1000) if host_rows_all is not None else None,
                              "rank_ms_per_step": {"min": min(per), "mean": sum(per) / len(per), "max": max(per)}}))
        finish(dist, rank)
        return

    import numpy as np
    from microaligner_amd import OptFlowRegistrator, Warper, synthetic
    from microaligner_amd.device import device_count, device_info, get_context

    ndev = device_count()
    if ndev < 1:
        raise RuntimeError("no HIP device: the benchmark has no CPU path")
    # one process per GPU; on a box with fewer GPUs than ranks (the 1-GPU test box) ranks share devices, which the
    # JSON line states (`devices`)
    dev_index = local_rank % ndev
    os.environ["MICROALIGNER_DEVICE"] = str(dev_index)   # what get_context() inside register()/warp() picks up
    # one process per GPU, next to its GPU: the CPUs of the device's NUMA node (page-locked buffers and the pages the input
    # synthesis first touches are then local to the GPU's PCIe root: SURVEY 8e, "per-rank affinity")
    from microaligner_amd.device import bind_to_device_numa, set_affinity
    all_cpus = sorted(os.sched_getaffinity(0))
    bound_cpus = bind_to_device_numa(dev_index)
    ctx = get_context()
    if args.no_companion:
        ctx.companion_stream = False

    np_dtype = np.uint8 if args.dtype == "u8" else np.float32
    freg = None
    if wl.get("affine") and args.feature_init:
        from microaligner_amd import FeatureRegistrator
        freg = FeatureRegistrator()
        freg.verbose = False
        freg.skip_repeated_rounds = not args.recompute_rounds
    # (ref, mov) device pairs of this rank; for the mosaic workload also the known inverse matrix and the host arrays
    # FeatureRegistrator takes
    work = []
    for idx in my_pairs:
        inv_affine, host = None, None
        if wl.get("affine"):
            # mosaic tile: cell-like texture, the moving image misplaced by a known similarity (rotation <= 0.5 deg,
            # shift <= 20 px) plus a smooth residual.  Affine initialisation: the known matrix (default; SURVEY 8d:
            # "ground-truth matrix where opencv-contrib is absent") or, with --feature-init, FeatureRegistrator.register()
            # inside the timed step (its sparse selection / matching glue runs on the host)
            ref, mov, M = synthetic.make_mosaic_tile(H, W, seed=1 + idx, dtype=np_dtype)
            inv_affine = np.vstack([M, [0, 0, 1]])    # pinv of the 3x3 of T = M^-1: what transform_img_with_tmat applies
            host = (ref, mov) if freg is not None else None
        else:
            ref, mov = synthetic.make_pair(H, W, seed=1 + idx, dtype=np_dtype)
        work.append((ctx.asdevice(ref), ctx.asdevice(mov), inv_affine, host))
    keep_host = world == 1 and not args.no_variants and not args.pairs_total
    want_modes = (args.host_modes == "on" or (args.host_modes == "auto" and world > 1)) and not wl.get("affine")
    if not keep_host and freg is None and not want_modes:
        ref = mov = None
    ctx.forget_host_arrays()

    reg = OptFlowRegistrator()
    reg.verbose = False
    reg.muladd_fused = args.fused
    reg.dog_muladd_fused = args.dog_fused
    for k, v in params.items():
        setattr(reg, k, v)
    warper = Warper()
    warper.tile_size, warper.overlap = reg.tile_size, reg.overlap

    def one_pair(dref, dmov, inv_affine, host):
        if freg is not None:
            # the pair is resident in HBM when the timed region starts (the metric's rule for every workload): the affine
            # initialisation reads the same device arrays the optical-flow stage reads; --feature-init-from-host times the
            # numpy entry instead (+ the upload of both images: 2 x 64 MB, ~3 ms)
            freg.ref_img, freg.mov_img = host if args.feature_init_from_host else (dref, dmov)
            t_mat = freg.register()
            m = ctx.warp_affine(dmov, np.linalg.pinv(np.vstack([t_mat, [0, 0, 1]])))
        else:
            m = dmov if inv_affine is None else ctx.warp_affine(dmov, inv_affine)   # transform_img_with_tmat (utils.py:98-114)
        reg.ref_img, reg.mov_img = dref, m
        flow = reg.register()
        warper.image, warper.flow = m, flow
        return flow, warper.warp()

    def step():
        out = None
        for item in work:
            out = one_pair(*item)
        return out

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # The timed loop keeps the previous step's results alive while the next step runs (`out = step()`), so it cycles
    # through TWO result buffers of each kind; a single warm-up step creates only one.  Put the second one into the
    # context's pool now: a fresh 2 GB hipMalloc inside the timed region costs ~170 ms (the driver clears new VRAM).
    spare = [ctx.empty((H, W, 2), np.float32) for _ in range(2)] + [ctx.empty((H, W), np_dtype) for _ in range(2)]
    del spare

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_reset()
    ctx.profile(True)
    t0 = time.perf_counter()
    marks = []
    for _ in range(args.steps):
        out = step()
        marks.append(time.perf_counter())   # host-side: register() synchronises once per level, the final warp does not
    ctx.sync()
    t1 = time.perf_counter()
    step_ms_host = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)]
    barrier()
    ctx.profile(False)
    elapsed = reduce_max(t1 - t0)
    # the shader clock the chip holds under the blur kernels' instruction mix, while it is still hot
    clock_ghz = ctx.clock_probe(20.0)
    # results of the sharded job come back to rank 0 over gloo (host side), timed apart from the device time: per pair a
    # checksum of the flow and of the warped image (the arrays themselves stay where the next pipeline stage needs them)
    tg0 = time.perf_counter()
    summary = None
    if out is not None:
        flow, warped = out
        summary = {"pair": my_pairs[-1] if my_pairs else None, "flow_minmax": [float(v) for v in ctx.minmax(flow)],
                   "warped_minmax": [float(v) for v in ctx.minmax(warped)]}
    info = device_info(dev_index)
    # the headline's rows are gathered BEFORE the host modes: nothing that happens in those can cost the line any more
    rows = rank_table(dist, world, rank, {
        "rank": rank, "device": dev_index, "pci_bus_id": info["pci_bus_id"], "name": info["name"],
        "hbm_free_gb": round(info["mem_free"] / 2 ** 30, 1), "hbm_total_gb": round(info["mem_total"] / 2 ** 30, 1),
        "cpus": (f"{cpu_ranges(bound_cpus)} ({len(bound_cpus)} of {len(all_cpus)}, local to the device)"
                 if bound_cpus else f"all {len(all_cpus)} (no NUMA binding: topology unknown or single node)"),
        "pairs": my_pairs, "ms_per_step": round((t1 - t0) / args.steps * 1e3, 3), "clock_ghz": round(clock_ghz, 3),
        "result": summary})
    gather_ms = (time.perf_counter() - tg0) * 1e3
    # ---- informational legs of a multi-rank run.  The headline's rows are on rank 0; nothing below enters a torch.distributed
    # collective after the one broadcast of rank 0's decisions: barriers and rows go through SoftSync, every wait bounded.
    host_rows_all, host_skip, shared_ms, shared_note, soft = None, None, None, None, None
    all_have_results = reduce_max(0.0 if out is not None else 1.0) < 0.5     # (--pairs-total below N leaves ranks without a pair)
    want_shared = world > 1 and all_have_results and not args.no_shared_results
    want_modes = want_modes and all_have_results
    plan = [None, None, 0]           # why the shared-results leg is skipped, why the host modes are, the store's port
    if rank == 0 and want_shared:
        need = world * (H * W * 8 + H * W * np.dtype(np_dtype).itemsize)
        try:
            st = os.statvfs("/dev/shm")
            room = st.f_bavail * st.f_frsize
        except OSError:
            room = 0
        if room <= 1.2 * need:
            plan[0] = f"skipped: /dev/shm has {room / 2 ** 30:.0f} GiB free, {need / 2 ** 30:.0f} GiB needed"
    if rank == 0 and want_modes:
        # every rank holds ~10 GiB of host arrays in these modes: do they fit beside each other
        need = host_modes_bytes(H, W, np.dtype(np_dtype).itemsize, args.host_mode_pages) * world
        room, how = available_host_bytes()
        if room is not None and need > 0.8 * room:
            plan[1] = f"skipped: {world} ranks need {need / 2 ** 30:.0f} GiB of host memory, {room / 2 ** 30:.0f} GiB available ({how})"
    if dist is not None:
        if rank == 0:
            plan[2] = free_port()
        dist.broadcast_object_list(plan, src=0)        # the last collective before the line
        try:
            soft = soft_sync(dist, world, rank, args.host_mode_timeout, plan[2])
        except Exception as e:   # noqa: BLE001
            print(f"bench.py: rank {rank}: store of the informational legs: {e!r}", file=sys.stderr)
    # the data plane of a multi-rank job: each rank's download engine writes its (last) flow and warped image straight into
    # a node-wide shared array (parallel.shared_array; the reference writes pages into its memmapped output,
    # __main__.py:116-132) -- nothing but reports crosses the control plane
    if want_shared and plan[0]:
        shared_note = plan[0]
    elif want_shared:
        try:
            from microaligner_amd import parallel
            if soft is None:
                raise RuntimeError("no store for the informational legs")
            sync = soft.section("shared results").barrier
            name = f"ma_bench_{os.environ.get('MASTER_PORT', '0')}"
            ts0 = time.perf_counter()
            # (unlink=True: the names are gone once every rank has mapped them -- nothing can be left in /dev/shm)
            flows = parallel.shared_array(name + "_flow", (world, H, W, 2), np.float32, unlink=True, barrier=sync)
            warps = parallel.shared_array(name + "_warp", (world, H, W), np_dtype, unlink=True, barrier=sync)
            out[0].numpy(out=flows[rank])
            out[1].numpy(out=warps[rank])
            sync()
            shared_ms = (time.perf_counter() - ts0) * 1e3
            if rank == 0:   # every rank's rows arrived: a strided sample per rank, next to what the rank reports over gloo
                shared_note = [float(flows[r][::997, ::991].sum()) for r in range(world)]
            del flows, warps
        except Exception as e:   # noqa: BLE001 -- informational: the line does not depend on it
            shared_ms, shared_note = None, f"failed: {e!r}"
    if want_modes and plan[1]:
        want_modes, host_skip = False, plan[1]
    if want_modes:
        del out
        out = None
        ctx.trim()        # the headline's pooled buffers go back: the stream needs its own slots and lanes
        host_rows_all = host_modes_exchange(
            soft, world, rank,
            lambda sync, rows: host_modes_measure(ctx, ref, mov, params, args.host_mode_pairs, args.host_mode_pages, sync, rows))
    out = None

    if rank == 0:
        prof = ctx.profile_get()
        iters, esz = reg.num_iterations, np.dtype(np_dtype).itemsize
        # with use_dog the Farneback inputs are the uint8 DOG images (1 B/px), not the level images
        fb_esz = 1 if reg.use_dog else esz
        pristine = (not args.size and not args.fused and not args.dog_fused and not args.no_dog and args.dtype == "f32"
                    and not args.pairs_total and not args.feature_init)
        tps, tsrc, tnote = pmc_traffic(args.workload) if pristine else ({}, None, "not the profiled command line")
        win = reg.overlap - (1 - reg.overlap % 2)
        nsteps_prof = args.steps * max(1, len(work))
        kclk, kclk_src = kernel_clocks(args.workload) if pristine else ({}, None)
        kernels = {k: roofline_entry(k, v, iters, fb_esz if k == "polyexp_m0" else esz, nsteps_prof, tps, win // 2, tsrc,
                                     clock_ghz, args.fused if k != "dog" else args.dog_fused, tnote, kclk.get(k), kclk_src)
                   for k, v in prof.items()}
        kernels = {k: v for k, v in kernels.items() if v}
        total_kernel_ms = sum(v["ms"] for v in prof.values())
        dominant = max(kernels, key=lambda k: prof[k]["ms"]) if kernels else None
        for k in kernels:
            kernels[k]["share_of_kernel_time"] = round(prof[k]["ms"] / total_kernel_ms, 4)
        per = [r["ms_per_step"] for r in rows]
        res = {
            "metric": METRIC, "value": round(pairs_per_step * H * W * args.steps / elapsed / 1e6, 2), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if args.pairs_total else "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['desc']}" + (f" (size overridden to {H})" if args.size else ""),
                       "pairs_per_step": pairs_per_step, "tile_size": reg.tile_size, "overlap": reg.overlap,
                       "num_iterations": reg.num_iterations, "muladd": "fma" if args.fused else "mul+add",
                       "dog_muladd": "fma" if args.dog_fused else "mul+add",
                       "levels": [[r.factor, r.accepted] for r in reg.level_reports],
                       "devices": ndev, "affine_init": (("FeatureRegistrator on the numpy pair (upload timed)" if args.feature_init_from_host else
                                                         "FeatureRegistrator on the device-resident pair") if freg is not None else
                                                        "known matrix" if wl.get("affine") else None),
                       "feature_rounds": (("a round after a rejected one is REPLAYED (same images, deterministic steps: same outcome), "
                                           "not recomputed" if freg.skip_repeated_rounds else "every round recomputed")
                                          if freg is not None else None),
                       "parallelism": f"{pairs_per_step} independent pairs per step dealt round-robin to {world} rank(s), "
                                      f"one rank per GPU, {min(world, ndev)} GPU(s), no collective on the data path"},
            # one compact row per kernel group; the full entries of the dominant kernel (`roofline`), the north-star kernel
            # (`roofline_polyexp`) and the vertical pass (`roofline_blur_v`) close the line, where a tail reader sees them
            "kernels": {k: {f: v[f] for f in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches",
                                              "share_of_kernel_time", "px_per_launch") if f in v}
                        for k, v in kernels.items()},
            "kernel_time_ms_per_step": round(total_kernel_ms / args.steps, 3),
            "kernel_time_note": "sum of the launch durations on BOTH streams of the context: the dog() of the reference and "
                                "the moving image of every level run on a low-priority companion stream under the level "
                                "loop (register.hip), so the sum exceeds the step and launches that share the chip are "
                                "longer than they would be alone",
            "step_ms_host": step_ms_host,
            "sustained_clock_ghz": round(clock_ghz, 3),
            "library": ctx.lib.ma_version().decode(),
            "ranks": rows,
            "rank_ms_per_step": {"min": min(per), "mean": round(sum(per) / len(per), 3), "max": max(per)},
            "gather_ms": round(gather_ms, 3),
            "results_to_shared_array_ms": round(shared_ms, 1) if shared_ms is not None else None,
            "results_to_shared_array": shared_note,
        }
        if host_skip:
            res["host_modes"] = {"skipped": host_skip}
        if host_rows_all is not None:
            res["host_modes"] = host_modes_report(host_rows_all, H, W)
        def informational(store, name, fn):
            """An informational leg never costs the headline line: a failure is recorded in its place."""
            try:
                store[name] = fn()
            except Exception as e:   # noqa: BLE001
                import traceback
                store[name] = {"error": repr(e), "where": traceback.format_exc().strip().splitlines()[-3:]}

        if keep_host and not args.fused:
            V = res["variants"] = {}
            dref, dmov, inv_affine, _ = work[0]

            def leg_host_inclusive():
                # the drop-in API as the reference's callers use it, numpy in -> numpy out (PCIe inclusive)
                th = host_inclusive_leg(max(1, min(args.steps, 3)), ref, mov if inv_affine is None else
                                        ctx.warp_affine(dmov, inv_affine).numpy(), params)
                return {"value": round(H * W / th / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(th * 1e3, 3),
                        "what": "numpy in -> numpy out, ONE pair, the reference's statements: H2D of ref and mov, register(), "
                                "D2H of the flow, Warper.warp(mov, flow) (the flow register() returned is recognised as "
                                "resident; the caller's writable mov array is uploaded again), D2H of the warped image; "
                                "nothing overlaps: see host_stream for the sustained rate"}

            informational(V, "host_inclusive", leg_host_inclusive)
            if args.stream_pairs > 0 and inv_affine is None:
                # the sustained numpy -> numpy rate over a stream of distinct pairs (PCIe inclusive; never the headline
                # value), in the workload's dtype and in the pipeline-faithful uint8
                informational(V, "host_stream", lambda: host_stream_leg(args.stream_pairs, ref, mov, params, np_dtype))
                if np_dtype != np.uint8:
                    informational(V, "host_stream_u8", lambda: host_stream_leg(args.stream_pairs, ref, mov, params, np.uint8))
            if freg is None:
                del ref, mov

            def timed_steps():
                step()
                ctx.sync()
                tf0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                ctx.sync()
                return (time.perf_counter() - tf0) / args.steps

            def leg_muladd_fma():
                # the same workload with the window blur in the FMA rounding model (MA_FB_MULADD_FUSED: OpenCV builds whose
                # v_muladd is a fused multiply-add)
                reg.muladd_fused = True
                try:
                    tf = timed_steps()
                finally:
                    reg.muladd_fused = False
                return {"value": round(H * W / tf / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(tf * 1e3, 3)}

            def leg_dog_fma():
                # the dog() chain in the rounding model of OpenCV's AVX2 + FMA3 objects
                reg.dog_muladd_fused = True
                try:
                    td = timed_steps()
                finally:
                    reg.dog_muladd_fused = False
                return {"value": round(H * W / td / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(td * 1e3, 3),
                        "what": "dog() with fused multiply-adds (MA_DOG_FUSED_BLUR | MA_DOG_FUSED_SCALE)"}

            def leg_lanes():
                # `lanes` independent pairs in flight on this GPU, one context (HIP stream, workspace) and one host thread
                # per lane -- what parallel.register_pairs(lanes=...) does for a list of device-resident pairs
                tl = lanes_leg(args.lanes, args.steps, ctx.device, dref, dmov, params, reg.tile_size, reg.overlap)
                return {"value": round(H * W / tl / 1e6, 2), "unit": "Mpix/s", "ms_per_step": round(tl * 1e3, 3),
                        "pairs_in_flight": args.lanes}

            informational(V, "muladd_fma", leg_muladd_fma)
            if reg.use_dog and not args.dog_fused:
                informational(V, "dog_fma", leg_dog_fma)
            if args.page_warps > 0:
                # before the lanes leg: that one closes its contexts, and the driver clears released VRAM on the DMA engines
                # -- for about a second per 25 GiB every hipMemcpy runs at 30 instead of 57 GB/s (profiles/r04_notes.md,
                # tools/ubench_d2h_after_work.hip).  A pipeline keeps its contexts and pooled buffers and does not see that.
                informational(V, "page_warp", lambda: page_warp_leg(args.page_warps, H, W, reg.tile_size, reg.overlap))
            if args.lanes > 1 and inv_affine is None:
                informational(V, f"lanes{args.lanes}", leg_lanes)
        if world == 1 and freg is not None and not args.no_variants and freg.skip_repeated_rounds:
            def leg_rounds_recomputed():
                # the same tiles with every round of FeatureRegistrator recomputed, as the reference does (a round after a
                # rejected one sees the same images and finds the same answer; the default replays it)
                freg.skip_repeated_rounds = False
                try:
                    step()
                    ctx.sync()
                    tr0 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    ctx.sync()
                    tr = (time.perf_counter() - tr0) / (args.steps * max(1, len(work)))
                finally:
                    freg.skip_repeated_rounds = True
                return {"ms_per_tile": round(tr * 1e3, 3), "what": "FeatureRegistrator.skip_repeated_rounds = False: repeated rounds "
                        "recomputed instead of replayed; same matrices (tests/test_feature_reg.py::test_replayed_rounds_equal_recomputed_rounds)"}
            informational(res.setdefault("variants", {}), "feature_rounds_recomputed", leg_rounds_recomputed)
        if world == 1 and wl.get("affine") and not args.no_variants and args.lanes > 1 and len(work) > 1:
            def leg_tile_lanes():
                tl, n = tile_lanes_leg(args.lanes, args.steps, ctx.device, work, params, freg is not None,
                                        freg.skip_repeated_rounds if freg is not None else True)
                return {"ms_per_tile": round(tl * 1e3, 3), "value": round(H * W / tl / 1e6, 2), "unit": "Mpix/s", "tiles_in_flight": n,
                        "what": "the same tiles, the same statements, `tiles_in_flight` of them at a time on one GPU (one context "
                                "and host thread each): what parallel.align_pairs(lanes=...) does; the headline runs them one "
                                "after the other"}
            informational(res.setdefault("variants", {}), f"tile_lanes{args.lanes}", leg_tile_lanes)
        if world == 1 and not args.no_cpu_baseline and not args.pairs_total:
            set_affinity(all_cpus)      # the CPU baseline uses every core of the host, not just the GPU's node
            # a bounded sample: the full workload where >= 128 CPUs can really run, else 8192^2 (the GPU boxes of this pool grant
            # 16 CPUs of time: ~9 s there; the full 16384^2 pair would take most of a minute)
            sample = args.cpu_sample or (H if effective_cpus()[0] >= 128 else 8192)
            informational(res, "cpu_baseline", lambda: cpu_baseline(min(sample, H), params))
        for key, name in (("roofline_blur_v", "blur_v"), ("roofline_polyexp", "polyexp_m0")):
            if name != dominant:
                res[key] = kernels.get(name)
        if dominant != "blur_h_solve":
            res["roofline_blur_h_solve"] = kernels.get("blur_h_solve")
        res["roofline"] = kernels.get(dominant)
        print(json.dumps(res))
    finish(dist, rank)


if __name__ == "__main__":
    main()
