"""Multi-GPU driver: independent units, one process per HIP device, no data-path collective.

The hot path shards over independent units: (ref, mov) pairs -- cycles against a fixed
reference, channels, mosaic tiles -- and the channel x z pages that reuse one flow.  Nothing is exchanged
between units, so the partition is a static round-robin over ranks and the only communication is the optional
gather of results onto rank 0 through the host (torch.distributed with the gloo backend; the GPUs never talk
to each other, xGMI/RCCL are not involved).  Launch with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 your_script.py

each process picks its device from LOCAL_RANK (microaligner_amd.device.default_device).
The reference's cycle CHAIN (ref_{k+1} = warp(mov_k), __main__.py:418-424) is serial across cycles and is not
sharded here; its per-cycle page warps are (warp_pages).
"""
import os
from typing import Callable, List, Optional, Sequence


def world():
    """(rank, world_size) from torch.distributed if initialised, else from the torchrun environment, else (0, 1)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard(n_units: int, rank: int, world_size: int) -> List[int]:
    """Indices of the units rank `rank` owns: static round-robin, as dask would deal tasks to workers."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of size {world_size}")
    return list(range(rank, n_units, world_size))


_lane_contexts = {}   # (device, lane index) -> Context, kept for the life of the process (release_lane_contexts())


def _lane_context(k: int):
    """The context (own HIP stream, workspace, buffer pools) of lane k on this process's device.  Lane 0 is the
    process-wide context; the others are created on first use and kept: a context's workspace (tens of GB for
    16384^2 pairs) and its page-locked result buffers are expensive to set up, and callers come back."""
    from .device import Context, get_context
    base = get_context()
    if k == 0:
        return base
    key = (base.device, k)
    ctx = _lane_contexts.get(key)
    if ctx is None or ctx._closed:
        ctx = _lane_contexts[key] = Context(base.device)
    return ctx


def release_lane_contexts():
    """Close the cached lane contexts (their HBM workspaces and page-locked buffers go back to the driver)."""
    for key in list(_lane_contexts):
        _lane_contexts.pop(key).close()


def _unit(units: Sequence, i: int):
    """Unit i, materialised HERE: a unit may be a loader -- a zero-argument callable that reads or builds the inputs
    (pages of a TIFF, a crop of a memmap) -- so that only the rank that owns it ever touches its data (SURVEY 8e:
    "inputs read/generated in the worker")."""
    u = units[i]
    return u() if callable(u) else u


def _run_local(units: Sequence, mine: List[int], fn: Callable, lanes: int):
    """fn over this rank's units; with lanes > 1, `lanes` threads each drive their own context on the rank's
    device and pull units from a shared queue, so that several units are in flight on the GPU at once (the
    kernels of one unit fill the launch tails and host round trips of another: +3..5 % on 16384^2 pairs)."""
    if lanes <= 1 or len(mine) <= 1:
        return {i: fn(_unit(units, i)) for i in mine}
    import threading
    from .device import use_context
    todo, lock, results, errors = list(reversed(mine)), threading.Lock(), {}, []

    nl = min(lanes, len(mine))
    ctxs = [_lane_context(k) for k in range(nl)]   # created here, on the calling thread, one after the other

    def worker(k):
        try:
            with use_context(ctxs[k]):
                while not errors:
                    with lock:
                        if not todo:
                            break
                        i = todo.pop()
                    results[i] = fn(_unit(units, i))
                ctxs[k].sync()
        except BaseException as e:  # surfaced on the calling thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,), name=f"ma-lane-{k}") for k in range(nl)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results


def shared_array(name: str, shape, dtype, directory: str = "/dev/shm", unlink: bool = False, page_locked=True, barrier=None):
    """A node-wide array every rank maps: rank 0 creates `directory/name` (POSIX shared memory by default; any path
    works, e.g. next to the output TIFF), the others open it after a barrier.  COLLECTIVE (every rank calls it with
    the same arguments).  This is where results of a sharded run go instead of being pickled through the control plane:
    each rank's download engine writes its rows in place (run_sharded / register_pairs `out=`), the way the reference
    writes every page straight into its memmapped output (__main__.py:116-132).
    unlink=True: the name is removed as soon as every rank has mapped the file -- the mappings stay valid, the memory goes
    back when the last rank drops its array or dies, and nothing is left behind in /dev/shm whatever happens later (use it
    for scratch results; keep the name, and remove it with shared_array_unlink, when another process is to open it).
    page_locked=True: every rank page-locks, in ITS mapping, the rows of the leading axis that `shard()` deals to it (rank,
    rank + world, ...: the rows it will write; pinning the whole array in every rank would pin every page world-size
    times over) when it has a HIP device (device.host_register), so that its download engine writes results by DMA straight
    into the shared memory -- no staging copy: per result byte one pass over host DRAM instead of three, which is what keeps
    eight ranks within the host's memory bandwidth.  page_locked="all": the whole array (a rank that
    writes rows of its own choosing).  Where the runtime refuses (no device: gloo tests, --dry-run; a file on disk) the
    array is pageable as before; `arr_is_page_locked(arr)` tells.  MICROALIGNER_SHARED_PAGE_LOCK=0 in the environment turns
    the default off (hosts with a memlock limit: pinning makes a sparse /dev/shm file resident).
    barrier: the callable the ranks meet at instead of torch.distributed's barrier (a caller with a bounded-wait barrier of its
    own, bench.py's informational legs: a rank that fails in here must not hold the others for the process group's timeout)."""
    import numpy as np
    rank, ws = world()
    barrier = barrier or _barrier
    if page_locked is True and os.environ.get("MICROALIGNER_SHARED_PAGE_LOCK", "1") == "0":
        page_locked = False       # hosts with a memlock limit: the default can be switched off from outside
    # SINGLE NODE: the file is created once and every rank maps that one file.  A launch that spans nodes has ranks whose
    # /dev/shm is another machine's -- refuse it rather than fail in np.load after the barrier.
    local_ws = int(os.environ.get("LOCAL_WORLD_SIZE", ws))
    if local_ws != ws:
        raise RuntimeError(f"shared_array is node-wide: WORLD_SIZE {ws} spans more than this node's {local_ws} ranks")
    path = os.path.join(directory, name)
    shape = tuple(int(v) for v in shape)
    if rank == 0:
        arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.dtype(dtype), shape=shape)
        arr.flush()
    barrier()
    if rank != 0:
        arr = np.load(path, mmap_mode="r+")
        if arr.shape != shape or arr.dtype != np.dtype(dtype):
            raise ValueError(f"{path}: found {arr.dtype}{arr.shape}, expected {np.dtype(dtype)}{shape}")
    if unlink:
        barrier()
        if rank == 0:
            os.unlink(path)
    if page_locked == "all" or (page_locked and (ws == 1 or arr.ndim < 2)):
        _page_lock(arr, [arr])
    elif page_locked and arr.shape[0] and arr[0].nbytes >= (1 << 20):
        # (rows of less than 1 MiB are not units of work: thousands of small registrations would cost more than they save --
        # such an array is page-locked as a whole on request, page_locked="all")
        _page_lock(arr, [arr[i] for i in range(rank, arr.shape[0], ws)])
    return arr


_PAGE_LOCKED = {}      # id(memmap) -> True, dropped with the array


def _page_lock(arr, pieces):
    """Best effort: only where a device exists in this process (the registration needs the HIP runtime).  pieces: views of
    `arr` (contiguous blocks) to register; the registrations end with `arr` (the views' finalizers hang on its buffer).
    All or nothing: when one piece cannot be registered the ones before it are released again, so that
    arr_is_page_locked() never says False about an array that is partly pinned."""
    try:
        from . import device
        if device.device_count() <= 0 or not pieces:
            return False
        done = []
        for p in pieces:
            if not device.host_register(p):
                for q in done:
                    device.host_unregister(q)
                return False
            done.append(p)
    except Exception:   # noqa: BLE001 -- no library / no device: the staging path serves the array
        return False
    import weakref
    _PAGE_LOCKED[id(arr)] = True
    weakref.finalize(arr, _PAGE_LOCKED.pop, id(arr), None)
    return True


def arr_is_page_locked(arr) -> bool:
    """True when shared_array page-locked this array's memory in this process."""
    return bool(_PAGE_LOCKED.get(id(arr)))


def shared_array_unlink(name: str, directory: str = "/dev/shm"):
    """Remove a shared_array's backing file (collective: rank 0 unlinks after everybody is done with it)."""
    _barrier()
    if world()[0] == 0:
        try:
            os.unlink(os.path.join(directory, name))
        except FileNotFoundError:
            pass


def _barrier():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
    except ImportError:
        pass


def _store(out, i, res):
    """Write unit i's result into the caller's array(s): out is one array (n_units, ...) or a tuple of them, matching a
    result that is one array or a tuple (None entries are skipped)."""
    import numpy as np
    outs = out if isinstance(out, (tuple, list)) else (out,)
    ress = res if isinstance(res, (tuple, list)) else (res,)
    if len(outs) != len(ress):
        raise ValueError(f"unit {i}: {len(ress)} result arrays for {len(outs)} output arrays")
    for o, r in zip(outs, ress):
        if o is None or r is None:
            continue
        if r is not o[i] and not np.shares_memory(r, o[i]):
            o[i][...] = r


def run_sharded(units: Sequence, fn: Callable, gather: bool = True, dst: int = 0, lanes: int = 1, out=None):
    """Apply `fn(unit)` to this rank's share of `units` (units that are callables are loaders: evaluated in the owning
    rank only).

    gather=False: returns {unit_index: result} for the local share (results stay where they were computed).
    gather=True : rank `dst` returns the full list in unit order, the other ranks return None; results travel
                  as host objects (numpy arrays) over gloo -- fine for reports and small arrays.
    out         : array (n_units, ...) or tuple of arrays that EVERY rank can write (shared_array, np.memmap): each
                  rank stores its results in place, nothing but a barrier crosses the control plane, and every rank
                  returns `out`.  The way to return images and flows from a multi-GPU run.
    lanes       : units kept in flight per GPU (threads with their own context); units and results must then be
                  host objects, because a DeviceArray is ordered on the stream of the context that made it.
    """
    rank, ws = world()
    mine = shard(len(units), rank, ws)
    local = _run_local(units, mine, fn, lanes)
    if out is not None:
        for i, res in local.items():
            _store(out, i, res)
        for o in (out if isinstance(out, (tuple, list)) else (out,)):
            if hasattr(o, "flush"):
                o.flush()
        _barrier()
        return out
    return _gather(local, len(units), gather, dst)


class PairResult:
    """What stream_pairs yields per pair: the host arrays, the per-level reports and the pair's position in the input."""

    __slots__ = ("index", "flow", "warped", "reports", "extra")

    def __init__(self, index, flow, warped, reports, extra=None):
        self.index, self.flow, self.warped, self.reports, self.extra = index, flow, warped, reports, extra

    def __iter__(self):          # `flow, warped = result`
        return iter((self.flow, self.warped))


def _optflow_stage(params, warp):
    """The compute stage of stream_pairs: register() (+ warp()) on device arrays -> (device results, reports, extra)."""
    from . import OptFlowRegistrator

    def stage(ctx, dref, dmov):
        reg = OptFlowRegistrator()
        reg.verbose = False
        for k, v in params.items():
            setattr(reg, k, v)
        reg.ref_img, reg.mov_img = dref, dmov
        flow = reg.register()
        warped = ctx.warp(dmov, flow, reg.tile_size, reg.overlap) if warp else None
        return [flow, warped], reg.level_reports, None

    return stage


def stream_pairs(pairs, params: Optional[dict] = None, warp: bool = True, depth: int = 2, stats: Optional[dict] = None,
                 stage: Optional[Callable] = None, out: Optional[Callable] = None, compute_lanes: Optional[int] = None):
    """Register a STREAM of (ref, mov) host pairs on this process's GPU, numpy in -> numpy out, with the upload of
    pair k+1 and the download of pair k-1 hidden behind the kernels of pair k.

    The reference meets its inputs one page at a time (TIFF pages read in __main__.py:398-433, "one page in memory",
    README.md:5) and overlaps work through dask (flow_calc.py:88-98, utils.py:117-123).  Here one context drives three
    engines (include/microaligner_hip.h, "transfer engines"), each from its own persistent host thread (the compute engine
    may have more than one lane, see compute_lanes):

      upload   : pulls the next pair from `pairs` (any iterable, evaluated lazily and on this thread: a generator that
                 reads files overlaps too), copies it into a free device input slot on the H2D stream, records an event;
      compute  : waits for that event on the compute stream, runs OptFlowRegistrator.register() and Warper.warp() on
                 the device arrays (the C engine, ma_optflow_register), records an event;
      download : waits for that event on the D2H stream, copies flow and warped image into page-locked host arrays
                 (the context's pool; pageable beyond its limit), hands the input slot back and yields.

    Events order the engines on the device; the host threads never wait for each other's copies.  Every byte crosses the
    bus once (ma_ctx_transfer_stats).  Results come out in input order as PairResult(index, flow, warped, reports);
    `for flow, warped in stream_pairs(...)` works too.  `depth`: pairs that may wait between two engines (device memory:
    depth + compute_lanes input slots, result buffers of as many pairs).  `stats`, if given, receives the busy time of each
    engine, the wall time and the byte counts when the stream ends.  `out`: optional callable index -> (flow_out,
    warped_out) host arrays to fill instead of pool arrays (e.g. rows of a memmap; either may be None).
    `stage`: replaces the compute step (align_pairs uses it): callable (ctx, dref, dmov) -> ([device arrays to
    download], reports, extra).  Bit-identical to the one-pair path: same kernels on the same inputs.
    `compute_lanes`: pairs registered at the same time, each by a compute thread with a context of its own (own stream,
    workspace and buffer cache: ~50 GB of HBM per lane at 16384^2); the coarse levels of one pair, whose few windows leave
    most of the chip idle, then run under the full-resolution level of another: 74.6 instead of 79.0 ms per 16384^2 pair
    with two lanes, 0.48 instead of 1.1 ms per 512^2 pair (three lanes: ~73 ms, the download engine comes into play;
    profiles/r04_notes.md).  Default (None): two lanes when two such working sets
    fit comfortably into the device's memory (judged from the first pair: 200 bytes per pixel and lane against 60 % of the
    HBM), else one.  Uploads, downloads and the order of the results are the same for any number of lanes."""
    import queue
    import threading
    import time

    import numpy as np

    from . import _lib as L
    from .device import _dt, get_context, use_context

    ctx = get_context()
    params = dict(params or {})
    if depth < 1:
        raise ValueError("depth must be >= 1")
    auto_lanes = compute_lanes is None
    if auto_lanes:
        compute_lanes = 2
    if compute_lanes < 1:
        raise ValueError("compute_lanes must be >= 1")
    stage = stage or _optflow_stage(params, warp)
    try:
        lane_ctxs = [ctx] + [_lane_context(k) for k in range(1, compute_lanes)]
    except Exception:
        if not auto_lanes:
            raise
        compute_lanes, lane_ctxs = 1, [ctx]          # no second context to be had: one lane
    effective = [compute_lanes]                        # lanes actually used: settled by the uploader at the first pair
    lanes_ready = threading.Event()

    def settle_lanes(first_pair_px):
        if auto_lanes and compute_lanes > 1 and first_pair_px is not None:
            try:
                from .device import device_info
                total = device_info(ctx.device)["mem_total"]
            except Exception:
                total = 0
            if 200.0 * first_pair_px * compute_lanes > 0.6 * total:
                effective[0] = 1
        lanes_ready.set()

    n_slots = depth + compute_lanes
    # host result arrays alive at once: one being filled, one queued, one just yielded, one the consumer still names
    n_results = depth + 1 + compute_lanes
    q_up, q_done, q_out = queue.Queue(depth), queue.Queue(depth + compute_lanes), queue.Queue(1)
    q_free = queue.Queue()
    stop = threading.Event()
    errors: list = []
    busy = {"h2d_busy_ms": 0.0, "d2h_busy_ms": 0.0, "compute_busy_ms": 0.0, "pairs": 0}
    reserved = set()
    timeline = {}                  # pair index -> host-side time stamps (seconds since the stream started) per engine
    END = object()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                pass
        return False

    def get(q):
        while not stop.is_set():
            try:
                return q.get(timeout=0.1)
            except queue.Empty:
                pass
        return END

    def fail(e):
        errors.append(e)
        stop.set()

    for c in lane_ctxs:
        c.sync()                   # everything enqueued so far is complete: any pooled buffer may be written by any engine
    up0, down0 = ctx.transfer_stats()

    class Slot:
        def __init__(self):
            self.ref = self.mov = None
            self.ev_up, self.ev_start, self.ev_done = ctx.event(), ctx.event(), ctx.event()

    slots = [Slot() for _ in range(n_slots)]

    def uploader():
        try:
            in_flight_shape = None
            for index, pair in enumerate(pairs):
                if stop.is_set():
                    return
                ref, mov = pair
                ref, mov = np.ascontiguousarray(ref), np.ascontiguousarray(mov)
                _dt(ref.dtype), _dt(mov.dtype)
                if ref.ndim != 2 or mov.ndim != 2:
                    raise ValueError(f"pair {index}: images must be 2-D, got {ref.shape} and {mov.shape}")
                if not lanes_ready.is_set():
                    settle_lanes(ref.size)
                key = (ref.shape, ref.dtype, mov.shape, mov.dtype)
                if key != in_flight_shape:
                    # (re)size the input slots: only when nothing is in flight, so that no engine can still be using a
                    # buffer the pool hands out (first pair: the pipeline is empty by construction)
                    have = []
                    while len(have) < n_slots:
                        s = get(q_free) if in_flight_shape is not None else slots[len(have)]
                        if s is END:
                            return
                        have.append(s)
                    for s in have:
                        s.ref = s.mov = None
                    if in_flight_shape is not None:
                        ctx.sync()
                    for s in have:
                        s.ref, s.mov = ctx.empty(ref.shape, ref.dtype), ctx.empty(mov.shape, mov.dtype)
                        q_free.put(s)
                    in_flight_shape = key
                slot = get(q_free)
                if slot is END:
                    return
                t0 = time.perf_counter()
                ctx.engine_upload(slot.ref, ref)
                ctx.engine_upload(slot.mov, mov)
                t1 = time.perf_counter()
                busy["h2d_busy_ms"] += (t1 - t0) * 1e3
                timeline[index] = {"h2d": (t0 - t_wall, t1 - t_wall)}
                ctx.engine_record(L.MA_ENGINE_H2D, slot.ev_up)
                if not put(q_up, (index, slot)):
                    return
            if not lanes_ready.is_set():
                settle_lanes(None)                     # an empty stream
            for _ in range(effective[0]):
                put(q_up, END)
        except BaseException as e:
            lanes_ready.set()
            fail(e)

    def computer(lane):
        lctx = lane_ctxs[lane]
        try:
            while not lanes_ready.wait(0.1):
                if stop.is_set():
                    return
            if lane >= effective[0]:                   # the working sets of this many lanes do not fit: this one stays idle
                put(q_done, END)
                return
            with use_context(lctx):
                while True:
                    item = get(q_up)
                    if item is END:
                        break
                    index, slot = item
                    t0 = time.perf_counter()
                    lctx.engine_wait(L.MA_ENGINE_COMPUTE, slot.ev_up)
                    lctx.engine_record(L.MA_ENGINE_COMPUTE, slot.ev_start)
                    devs, reports, extra = stage(lctx, slot.ref, slot.mov)
                    lctx.engine_record(L.MA_ENGINE_COMPUTE, slot.ev_done)
                    timeline[index]["compute_host"] = (t0 - t_wall, time.perf_counter() - t_wall)
                    timeline[index]["lane"] = lane
                    if not put(q_done, (index, slot, devs, reports, extra)):
                        return
            put(q_done, END)
        except BaseException as e:
            fail(e)

    def downloader():
        try:
            waiting, nxt, ends = {}, 0, 0       # results of the lanes arrive in any order and leave in input order
            while True:
                while nxt not in waiting and ends < compute_lanes:
                    item = get(q_done)
                    if item is END:
                        if stop.is_set():
                            return
                        ends += 1
                    else:
                        waiting[item[0]] = item
                if nxt not in waiting:
                    break
                item = waiting.pop(nxt)
                nxt += 1
                index, slot, devs, reports, extra = item
                ctx.engine_wait(L.MA_ENGINE_D2H, slot.ev_done)
                ctx.event_sync(slot.ev_done)               # the wait for the kernels is not transfer time
                t_done = time.perf_counter()
                gpu_ms = ctx.elapsed_ms(slot.ev_start, slot.ev_done)
                busy["compute_busy_ms"] += gpu_ms
                timeline[index]["compute_gpu_ms"] = gpu_ms
                timeline[index]["compute_done"] = t_done - t_wall
                put(q_free, slot)                          # the inputs are no longer read: the slot can be refilled
                targets = list(out(index)) if out is not None else [None] * len(devs)
                if len(targets) != len(devs):
                    raise ValueError(f"out({index}) must return {len(devs)} arrays (or None entries)")
                hosts = []
                t_busy = 0.0
                for k, (d, tgt) in enumerate(zip(devs, targets)):
                    if d is None:
                        hosts.append(None)
                        continue
                    if tgt is None:
                        if (d.shape, d.dtype) not in reserved and out is None:
                            ctx.host_reserve(d.shape, d.dtype, n_results)     # once per result shape, before its first copy
                            reserved.add((d.shape, d.dtype))
                        tgt = ctx.host_empty(d.shape, d.dtype, limit=n_results)
                    elif (tuple(tgt.shape) != d.shape or tgt.dtype != d.dtype or not tgt.flags.c_contiguous
                          or not tgt.flags.writeable):
                        raise ValueError(f"pair {index}: output {k} must be a writable C-contiguous {d.dtype} array of "
                                         f"shape {d.shape}")
                    t0 = time.perf_counter()
                    ctx.engine_download(d, tgt)
                    t_busy += time.perf_counter() - t0
                    hosts.append(tgt)
                busy["d2h_busy_ms"] += t_busy * 1e3
                busy["pairs"] += 1
                timeline[index]["d2h"] = (time.perf_counter() - t_wall - t_busy, time.perf_counter() - t_wall)
                del devs, item                             # device results go back to the pool: their copies are complete
                if not put(q_out, PairResult(index, hosts[0], hosts[1] if len(hosts) > 1 else None, reports, extra)):
                    return
            put(q_out, END)
        except BaseException as e:
            fail(e)

    threads = [threading.Thread(target=uploader, name="ma-engine-h2d", daemon=True),
               threading.Thread(target=downloader, name="ma-engine-d2h", daemon=True)]
    threads += [threading.Thread(target=computer, args=(k,), name=f"ma-engine-compute-{k}", daemon=True)
                for k in range(compute_lanes)]
    t_wall = time.perf_counter()
    for t in threads:
        t.start()
    try:
        while True:
            res = get(q_out)
            if res is END:
                break
            yield res
            res = None
        if errors:
            raise errors[0]
    finally:
        stop.set()
        for t in threads:
            t.join()
        try:
            for c in lane_ctxs:
                c.sync()
            for e in (L.MA_ENGINE_H2D, L.MA_ENGINE_D2H):
                ctx.engine_sync(e)
        finally:
            for s in slots:
                s.ref = s.mov = None
                for ev in (s.ev_up, s.ev_start, s.ev_done):
                    ctx.event_destroy(ev)
        if stats is not None:
            up1, down1 = ctx.transfer_stats()
            stats.update(busy, wall_ms=(time.perf_counter() - t_wall) * 1e3, h2d_bytes=up1 - up0, d2h_bytes=down1 - down0,
                         depth=depth, compute_lanes=effective[0], timeline=[timeline[i] for i in sorted(timeline)])


def register_pairs(pairs: Sequence, params: Optional[dict] = None, warp: bool = False, gather: bool = True,
                   lanes: int = 1, stream: Optional[bool] = None, out=None):
    """Register every (ref, mov) pair of `pairs` on this rank's GPU share.  Returns flows (and warped moving images if
    warp=True) as numpy arrays, in pair order on rank 0.
    A pair may be a LOADER: a zero-argument callable returning (ref, mov), evaluated in the owning rank only -- and, in
    stream mode, on the upload engine's thread, so that reading overlaps the kernels of the previous pair.
    stream (default: on for host pairs and loaders when lanes == 1): the rank's share goes through stream_pairs -- one
    context, transfers overlapped with the kernels -- instead of one blocking upload / compute / download per pair.
    out: (flows, warped) arrays with one row per pair -- shared_array() or np.memmap, the same on every rank; either may
    be None -- that the download engine fills in place; nothing is gathered then and every rank returns `out`.
    lanes > 1: the older scheme, `lanes` host threads each with a context of their own."""
    import numpy as np
    from . import OptFlowRegistrator, Warper
    params = dict(params or {})

    def one(pair):
        ref, mov = pair
        reg = OptFlowRegistrator()
        reg.verbose = False
        for k, v in params.items():
            setattr(reg, k, v)
        reg.ref_img, reg.mov_img = ref, mov
        flow = reg.register()
        if not warp:
            return flow
        w = Warper()
        w.tile_size, w.overlap = reg.tile_size, reg.overlap
        w.image, w.flow = mov, flow
        return flow, w.warp()

    rank, ws = world()
    mine = shard(len(pairs), rank, ws)
    host_in = all(callable(pairs[i]) or (isinstance(pairs[i][0], np.ndarray) and isinstance(pairs[i][1], np.ndarray))
                  for i in mine)
    if stream is None:
        stream = lanes <= 1 and host_in and (len(mine) > 1 or out is not None)
    if out is not None and (not isinstance(out, (tuple, list)) or len(out) != 2):
        raise ValueError("out must be a (flows, warped) pair of arrays with one row per pair (either may be None)")
    if not stream:
        if out is not None:
            return run_sharded(pairs, one, lanes=lanes, out=out if warp else out[0])
        return run_sharded(pairs, one, gather=gather, lanes=lanes)
    if not host_in:
        raise ValueError("stream=True needs host (numpy) pairs or loaders")
    local = {}
    sink = None
    if out is not None:
        def sink(k):                                   # rows of the caller's arrays for the k-th pair of this rank
            i = mine[k]
            return (out[0][i] if out[0] is not None else None, out[1][i] if (warp and out[1] is not None) else None)
    for res in stream_pairs((_unit(pairs, i) for i in mine), params, warp=warp, out=sink):
        local[mine[res.index]] = (res.flow, res.warped) if warp else res.flow
    if out is not None:
        for o in out:
            if o is not None and hasattr(o, "flush"):
                o.flush()
        _barrier()
        return out
    return _gather(local, len(pairs), gather)


def _gather(local: dict, n_units: int, gather: bool, dst: int = 0):
    """{unit index: result} of this rank -> the full list in unit order on rank `dst` (None elsewhere)."""
    rank, ws = world()
    if not gather:
        return local
    if ws == 1:
        return [local[i] for i in range(n_units)]
    import torch.distributed as dist
    bucket: Optional[list] = [None] * ws if rank == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(n_units)]


def warp_pages(pages: Sequence, flow, tile_size: int = 1000, overlap: int = 100, gather: bool = True, out=None):
    """Apply ONE flow to many pages (channels x z-planes of a cycle, __main__.py:288-302,427-433).  Pages are dealt
    round-robin to the ranks (a page may be a loader: a zero-argument callable evaluated by its owner only); each rank
    uploads the flow once and streams its share through the page-warp driver (ma_warp_pages_host: upload, warp and download
    of consecutive pages on the three engines of its context).
    out: an array (n_pages, H, W) every rank can write -- shared_array() or the np.memmap of the output file, the way the
    reference writes every page into its memmapped TIFF (__main__.py:116-132) -- whose rows the download engine fills in
    place; nothing is gathered then and every rank returns `out`.  Otherwise: the warped pages in page order on rank 0
    (gather=True, pickled over gloo: for small jobs) or {page_index: array} for the local share."""
    from .device import get_context
    rank, ws = world()
    mine = shard(len(pages), rank, ws)
    ctx = get_context()
    rows = [out[i] for i in mine] if out is not None else None
    outs = ctx.warp_pages([_unit(pages, i) for i in mine], ctx.asdevice(flow), tile_size, overlap, rows) if mine else []
    if out is not None:
        if hasattr(out, "flush"):
            out.flush()
        _barrier()
        return out
    return _gather(dict(zip(mine, outs)), len(pages), gather)


def register_cycle_chain(cycles: Sequence, ref_channel_ids=None, params: Optional[dict] = None):
    """The optical-flow cycle loop of the reference pipeline on in-memory stacks (__main__.py:398-433).

    cycles: sequence of (C, Z, H, W) arrays, one per cycle, first = reference cycle.  For every later cycle the
    max-projected, uint8-normalised reference channel (utils.py:75-95) is registered against the *previous cycle's
    warped* reference image (the chain of :418-424), and every channel x z page of the cycle is warped with that
    one flow (warp_and_save_pages, :288-302) -- flow resident in HBM, page transfers overlapped.
    Returns (aligned cycles as arrays of the input shape and dtype, flows as numpy arrays; flows[0] is None).
    The chain is serial across cycles; run it on one rank (pages of a cycle can be sharded with warp_pages)."""
    import numpy as np
    from . import OptFlowRegistrator, Warper
    from .shared_modules.utils import max_project_and_normalize
    params = dict(params or {})
    ref_channel_ids = list(ref_channel_ids) if ref_channel_ids is not None else [0] * len(cycles)
    if len(ref_channel_ids) != len(cycles):
        raise ValueError("one reference channel id per cycle is required")
    aligned, flows, ref_img = [], [], None
    for cyc, stack in enumerate(cycles):
        stack = np.asarray(stack)
        if stack.ndim != 4:
            raise ValueError(f"cycle {cyc}: expected a (C, Z, H, W) stack, got shape {stack.shape}")
        C_, Z_, H, W = stack.shape
        mov_img = max_project_and_normalize(stack[ref_channel_ids[cyc]], on_device=True)
        if cyc == 0:
            ref_img = mov_img                       # "Skipping as it is a reference image" (:411-413)
            aligned.append(stack.copy())
            flows.append(None)
            continue
        reg = OptFlowRegistrator()
        reg.verbose = False
        for k, v in params.items():
            setattr(reg, k, v)
        reg.ref_img, reg.mov_img = ref_img, mov_img
        flow = reg.register()
        w = Warper()
        w.tile_size, w.overlap = reg.tile_size, reg.overlap
        w.image, w.flow = mov_img, flow
        ref_img = w.warp()                          # reference of the next cycle (:424)
        w.flow = flow
        out = np.empty_like(stack)
        w.warp_pages([stack[c, z] for c in range(C_) for z in range(Z_)],
                     out=[out[c, z] for c in range(C_) for z in range(Z_)])
        aligned.append(out)
        flows.append(flow.numpy())
    return aligned, flows


def align_pairs(pairs: Sequence, feature_params: Optional[dict] = None, optflow_params: Optional[dict] = None,
                gather: bool = True, lanes: int = 1, stream: Optional[bool] = None):
    """Two-stage alignment of independent (ref, mov) pairs -- mosaic tiles, BASELINE cfg5 -- sharded over the ranks:
    feature-based affine initialisation (FeatureRegistrator, the pipeline's first stage, __main__.py:257-286),
    transform_img_with_tmat, then the optical-flow refinement and warp (OptFlowRegistrator + Warper, :398-433).
    Returns, per pair and in pair order on rank 0, (aligned moving image, 2x3 matrix, flow).
    stream (default: on for host pairs of equal shape when lanes == 1): the rank's share goes through stream_pairs, the
    two stages of pair k running on the device arrays while pair k+1 is uploaded and pair k-1 downloaded."""
    import numpy as np
    from . import FeatureRegistrator, OptFlowRegistrator, Warper, transform_img_with_tmat
    feature_params, optflow_params = dict(feature_params or {}), dict(optflow_params or {})

    def one(pair):
        ref, mov = pair
        freg = FeatureRegistrator()
        freg.verbose = False
        for k, v in feature_params.items():
            setattr(freg, k, v)
        freg.ref_img, freg.mov_img = ref, mov
        t_mat = freg.register()
        affine = transform_img_with_tmat(mov, ref.shape, t_mat)
        oreg = OptFlowRegistrator()
        oreg.verbose = False
        for k, v in optflow_params.items():
            setattr(oreg, k, v)
        oreg.ref_img, oreg.mov_img = ref, affine
        flow = oreg.register()
        w = Warper()
        w.tile_size, w.overlap = oreg.tile_size, oreg.overlap
        w.image, w.flow = affine, flow
        return w.warp(), t_mat, flow

    def stage(ctx, dref, dmov):
        # the same statements on device arrays (transform_img_with_tmat: utils.py:98-114 without the padding, the shapes
        # are equal here)
        freg = FeatureRegistrator()
        freg.verbose = False
        for k, v in feature_params.items():
            setattr(freg, k, v)
        freg.ref_img, freg.mov_img = dref, dmov
        t_mat = freg.register()
        identity = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
        if np.array_equal(t_mat, identity):
            affine = dmov
        else:
            affine = ctx.warp_affine(dmov, np.linalg.pinv(np.append(np.asarray(t_mat, dtype=np.float64), [[0, 0, 1]], axis=0)))
        oreg = OptFlowRegistrator()
        oreg.verbose = False
        for k, v in optflow_params.items():
            setattr(oreg, k, v)
        oreg.ref_img, oreg.mov_img = dref, affine
        flow = oreg.register()
        return [ctx.warp(affine, flow, oreg.tile_size, oreg.overlap), flow], oreg.level_reports, t_mat

    rank, ws = world()
    mine = shard(len(pairs), rank, ws)
    same = all(isinstance(pairs[i][0], np.ndarray) and isinstance(pairs[i][1], np.ndarray)
               and pairs[i][0].shape == pairs[i][1].shape and pairs[i][0].dtype == pairs[i][1].dtype for i in mine)
    if stream is None:
        stream = lanes <= 1 and same and len(mine) > 1
    if not stream:
        return run_sharded(pairs, one, gather=gather, lanes=lanes)
    if not same:
        raise ValueError("stream=True needs host (numpy) pairs whose two images have the same shape and dtype")
    local = {}
    for res in stream_pairs((pairs[i] for i in mine), stage=stage):
        local[mine[res.index]] = (res.flow, res.extra, res.warped)   # stage order: [warped image, flow]
    return _gather(local, len(pairs), gather)
