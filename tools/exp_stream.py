#!/usr/bin/env python3
"""Experiment: where the time of parallel.stream_pairs goes (run on the GPU box).

    python tools/exp_stream.py [--size 16384] [--pairs 6] [--dtype f32|u8] [--inputs pageable|pinned|both]

Prints, per variant, ms per pair (steady state and wall), the busy time of the three engines and the copy rates.
`pinned` inputs live in page-locked memory from the context's pool (what a loader that reads straight into
Context.host_empty() arrays would hand over); `pageable` inputs are ordinary numpy arrays.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--pairs", type=int, default=12)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--inputs", default="both")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--trim", action="store_true", help="after --first-single: hand every pooled buffer back first")
    ap.add_argument("--first-single", action="store_true", help="run the one-pair numpy path first (as bench.py does)")
    ap.add_argument("--lanes", type=int, default=1, help="compute lanes of stream_pairs")
    ap.add_argument("--no-bind", action="store_true", help="leave the process on every CPU of the host")
    args = ap.parse_args()
    from microaligner_amd import parallel, synthetic
    from microaligner_amd.device import bind_to_device_numa, device_info, get_context
    info = device_info(0)
    bound = [] if args.no_bind else bind_to_device_numa(0)
    print(f"device {info['name']!r} {info['pci_bus_id']}; bound to {len(bound)} local CPUs" if bound else
          f"device {info['name']!r} {info['pci_bus_id']}; not bound ({len(os.sched_getaffinity(0))} CPUs)", flush=True)
    ctx = get_context()
    dt = np.float32 if args.dtype == "f32" else np.uint8
    H = W = args.size
    params = dict(num_pyr_lvl=4, use_full_res_img=True, use_dog=True)
    if H < 4096:
        params = dict(num_pyr_lvl=2, use_full_res_img=True, use_dog=True, tile_size=500, overlap=50)
    ref, mov = synthetic.make_pair(H, W, 1, dt)
    base = [(np.roll(ref, 53 * k, axis=0), np.roll(mov, 53 * k, axis=0)) for k in range(args.pairs)]
    if args.first_single:
        # what bench.py's host_inclusive leg does before its stream leg: the one-pair drop-in path on numpy arrays
        from microaligner_amd import OptFlowRegistrator, Warper
        for _ in range(3):
            reg = OptFlowRegistrator()
            reg.verbose = False
            for k, v in params.items():
                setattr(reg, k, v)
            reg.ref_img, reg.mov_img = ref, mov
            flow = reg.register()
            w = Warper()
            w.image, w.flow = mov, flow
            w.warp()
        del flow
        if args.trim:
            ctx.forget_host_arrays()
            ctx.trim()
        print("ran the one-pair path three times first" + (", then trimmed the context" if args.trim else ""), flush=True)
    variants = ["pageable", "pinned"] if args.inputs == "both" else [args.inputs]
    for v in variants:
        if v == "pinned":
            pairs = []
            for a, b in base:
                pa, pb = ctx.host_empty(a.shape, a.dtype, limit=2 * args.pairs), ctx.host_empty(b.shape, b.dtype, limit=2 * args.pairs)
                pa[...] = a
                pb[...] = b
                pairs.append((pa, pb))
        else:
            pairs = base
        for _ in parallel.stream_pairs(pairs[:3], params, depth=args.depth, compute_lanes=args.lanes):
            pass
        stats, marks = {}, []
        t0 = time.perf_counter()
        for res in parallel.stream_pairs(pairs, params, depth=args.depth, stats=stats, compute_lanes=args.lanes):
            marks.append(time.perf_counter())
        wall = (marks[-1] - t0) * 1e3
        half, L_ = len(marks) // 2, max(1, stats["compute_lanes"])     # the later half, intervals one lane group apart, median
        spans = sorted((marks[i] - marks[i - L_]) / L_ for i in range(max(half, L_), len(marks)))
        steady = (spans[len(spans) // 2] if len(spans) % 2 else 0.5 * (spans[len(spans) // 2 - 1] + spans[len(spans) // 2])) * 1e3
        n = len(pairs)
        print(f"[{v:8s}] lanes {args.lanes} {args.dtype} {H}x{W}: steady {steady:7.2f} ms/pair, wall {wall / n:7.2f} ms/pair | busy per pair: "
              f"h2d {stats['h2d_busy_ms'] / n:6.1f} ms ({stats['h2d_bytes'] / stats['h2d_busy_ms'] / 1e6:5.1f} GB/s)  "
              f"compute {stats['compute_busy_ms'] / n:6.1f} ms  d2h {stats['d2h_busy_ms'] / n:6.1f} ms "
              f"({stats['d2h_bytes'] / stats['d2h_busy_ms'] / 1e6:5.1f} GB/s)", flush=True)
        if args.timeline:
            for i, t in enumerate(stats["timeline"]):
                print(f"   pair {i:2d}: h2d {t['h2d'][0] * 1e3:7.1f}-{t['h2d'][1] * 1e3:7.1f}  compute(host) "
                      f"{t['compute_host'][0] * 1e3:7.1f}-{t['compute_host'][1] * 1e3:7.1f} done {t['compute_done'] * 1e3:7.1f} "
                      f"(gpu {t['compute_gpu_ms']:6.1f})  d2h {t['d2h'][0] * 1e3:7.1f}-{t['d2h'][1] * 1e3:7.1f}  yield {(marks[i] - t0) * 1e3:7.1f}")
        del pairs


if __name__ == "__main__":
    main()
