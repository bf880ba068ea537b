"""Host-side budget of an N-rank run WITHOUT a device: can the host memory system feed N GPUs?

    python3 tools/host_soak.py --ranks 1,2,4,8 [--seconds 4] [--scale 1.0] [--mode stream|pages]

Every rank of a multi-GPU run is a process bound to the CPUs of its GPU's NUMA node (device.bind_to_device_numa) whose
transfer engines stage pageable numpy arrays through page-locked chunks with the library's two copy pools
(ma_host_parallel_copy: caller's array -> upload chunk; ma_host_stream_copy: download chunk -> caller's array, non-temporal
stores).  Those copies need no device.  This tool starts N such processes -- the ranks of a node dealt to its NUMA nodes as
an 8-GPU box deals its GPUs, half per socket -- each cycling, as fast as it can and in both directions at once (two
threads, as the engines run), the bytes ONE unit of the workload moves:

    stream : a cfg3 pair through parallel.stream_pairs   -- 2.15 GB in (ref + mov, f32 16384^2), 3.22 GB out (flow + warped)
    pages  : a 16384^2 uint16 page through the page-warp driver -- 0.54 GB in, 0.54 GB out

and prints, per rank count, the aggregate staging-copy rate against what the GPUs would ask for (units per second the
kernels sustain, measured on one GPU).  The DMA's own reads and writes of the chunks are NOT emulated: add one more
pass over the same bytes when reading the result against the host's DRAM bandwidth (the JSON says so).
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNK = 32 << 20
MODES = {
    # bytes in, bytes out, units per second and GPU the kernels sustain (bench.py, one MI355X)
    "stream": dict(bytes_in=2 * 16384 * 16384 * 4, bytes_out=3 * 16384 * 16384 * 4, units_per_s=1 / 0.075,
                   unit="cfg3 pair numpy -> numpy (stream_pairs)"),
    "pages": dict(bytes_in=16384 * 16384 * 2, bytes_out=16384 * 16384 * 2, units_per_s=21.3e9 / (16384 * 16384),
                  unit="16384^2 uint16 page (warp_pages)"),
}


def numa_nodes():
    """[[cpus of node 0], [cpus of node 1], ...] restricted to the CPUs this process may use."""
    allowed = os.sched_getaffinity(0)
    nodes = []
    base = "/sys/devices/system/node"
    try:
        names = sorted((d for d in os.listdir(base) if d.startswith("node") and d[4:].isdigit()), key=lambda d: int(d[4:]))
    except OSError:
        names = []
    for d in names:
        cpus = set()
        try:
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if not part:
                    continue
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
        except OSError:
            continue
        cpus &= allowed
        if cpus:
            nodes.append(sorted(cpus))
    return nodes or [sorted(allowed)]


def rank_main(rank, nranks, mode, scale, seconds, start_evt, q):
    import numpy as np
    nodes = numa_nodes()
    cpus = nodes[rank * len(nodes) // nranks]          # the ranks of a node, dealt to its NUMA nodes in blocks
    os.sched_setaffinity(0, cpus)                      # before the library creates its copy workers
    from microaligner_amd import _lib
    lib = _lib.load()
    m = MODES[mode]
    nin, nout = int(m["bytes_in"] * scale) // 4096 * 4096, int(m["bytes_out"] * scale) // 4096 * 4096
    src = np.ones(nin, np.uint8)                       # pageable, touched
    dst = np.ones(nout, np.uint8)
    ring_in = [np.ones(CHUNK, np.uint8) for _ in range(3)]
    ring_out = [np.ones(CHUNK, np.uint8) for _ in range(3)]
    done = {"in": 0, "out": 0}
    stop = threading.Event()

    def cycle(which, total, fn, big, ring, big_is_dst):
        k = 0
        while not stop.is_set():
            for off in range(0, total, CHUNK):
                n = min(CHUNK, total - off)
                b = big.ctypes.data + off
                r = ring[k % 3].ctypes.data
                fn(C.c_void_p(b if big_is_dst else r), C.c_void_p(r if big_is_dst else b), C.c_size_t(n))
                k += 1
                done[which] += n
                if stop.is_set():
                    break

    q.put(("ready", rank))
    start_evt.wait()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=cycle, args=("in", nin, lib.ma_host_parallel_copy, src, ring_in, False)),
          threading.Thread(target=cycle, args=("out", nout, lib.ma_host_stream_copy, dst, ring_out, True))]
    for t in ts:
        t.start()
    time.sleep(seconds)
    stop.set()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    q.put(("done", rank, done["in"] / dt, done["out"] / dt, len(cpus)))


def run(nranks, mode, scale, seconds):
    ctx = mp.get_context("spawn")
    q, start = ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=rank_main, args=(r, nranks, mode, scale, seconds, start, q)) for r in range(nranks)]
    for p in ps:
        p.start()
    for _ in ps:
        assert q.get(timeout=600)[0] == "ready"
    start.set()
    rows = sorted((q.get(timeout=seconds + 600) for _ in ps), key=lambda r: r[1])
    for p in ps:
        p.join()
    m = MODES[mode]
    gin, gout = sum(r[2] for r in rows) / 1e9, sum(r[3] for r in rows) / 1e9
    need_in = nranks * m["bytes_in"] * m["units_per_s"] / 1e9
    need_out = nranks * m["bytes_out"] * m["units_per_s"] / 1e9
    return {"ranks": nranks, "mode": mode, "unit": m["unit"],
            "staging_in_gb_s": round(gin, 1), "staging_out_gb_s": round(gout, 1),
            "needed_in_gb_s": round(need_in, 1), "needed_out_gb_s": round(need_out, 1),
            "covers": round(min(gin / need_in, gout / need_out), 2),
            "per_rank_in_gb_s": [round(r[2] / 1e9, 1) for r in rows], "per_rank_out_gb_s": [round(r[3] / 1e9, 1) for r in rows],
            "cpus_per_rank": rows[0][4],
            "dram_traffic_note": "each staged byte is read once and written once by these copies; the DMA engine reads (in) or "
                                 "writes (out) the chunk once more: DRAM traffic = 3 passes per staged byte, 1 pass for "
                                 "page-locked arrays that the DMA reaches in place"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--mode", default="stream", choices=sorted(MODES))
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the unit's bytes each rank cycles (memory of the host)")
    ap.add_argument("--seconds", type=float, default=4.0)
    args = ap.parse_args()
    print(json.dumps({"numa_nodes": [len(n) for n in numa_nodes()], "cpu_count": os.cpu_count()}))
    for n in (int(v) for v in args.ranks.split(",")):
        print(json.dumps(run(n, args.mode, args.scale, args.seconds)), flush=True)


if __name__ == "__main__":
    main()
