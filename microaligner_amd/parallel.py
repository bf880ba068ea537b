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


def _run_local(units: Sequence, mine: List[int], fn: Callable, lanes: int):
    """fn over this rank's units; with lanes > 1, `lanes` threads each drive their own context on the rank's
    device and pull units from a shared queue, so that several units are in flight on the GPU at once (the
    kernels of one unit fill the launch tails and host round trips of another: +3..5 % on 16384^2 pairs)."""
    if lanes <= 1 or len(mine) <= 1:
        return {i: fn(units[i]) for i in mine}
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
                    results[i] = fn(units[i])
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


def run_sharded(units: Sequence, fn: Callable, gather: bool = True, dst: int = 0, lanes: int = 1):
    """Apply `fn(unit)` to this rank's share of `units`.

    gather=False: returns {unit_index: result} for the local share (results stay where they were computed).
    gather=True : rank `dst` returns the full list in unit order, the other ranks return None; results travel
                  as host objects (numpy arrays) over gloo.
    lanes       : units kept in flight per GPU (threads with their own context); units and results must then be
                  host objects, because a DeviceArray is ordered on the stream of the context that made it.
    """
    rank, ws = world()
    mine = shard(len(units), rank, ws)
    local = _run_local(units, mine, fn, lanes)
    if not gather:
        return local
    if ws == 1:
        return [local[i] for i in range(len(units))]
    import torch.distributed as dist
    bucket: Optional[list] = [None] * ws if rank == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if rank != dst:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(len(units))]


def register_pairs(pairs: Sequence, params: Optional[dict] = None, warp: bool = False, gather: bool = True,
                   lanes: int = 1):
    """Register every (ref, mov) pair of `pairs` on this rank's GPU share, `lanes` pairs in flight per GPU.
    Returns flows (and warped moving images if warp=True) as numpy arrays, in pair order on rank 0."""
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

    return run_sharded(pairs, one, gather=gather, lanes=lanes)


def warp_pages(pages: Sequence, flow, tile_size: int = 1000, overlap: int = 100, gather: bool = True):
    """Apply ONE flow to many pages (channels x z-planes of a cycle, __main__.py:288-302,427-433).  Pages are dealt
    round-robin to the ranks; each rank uploads the flow once and streams its share through the overlapped
    page-warp driver (ma_warp_pages_host).  Returns the warped pages in page order on rank 0 (gather=True) or
    {page_index: array} for the local share."""
    from .device import get_context
    rank, ws = world()
    mine = shard(len(pages), rank, ws)
    ctx = get_context()
    outs = ctx.warp_pages([pages[i] for i in mine], ctx.asdevice(flow), tile_size, overlap) if mine else []
    local = dict(zip(mine, outs))
    if not gather:
        return local
    if ws == 1:
        return [local[i] for i in range(len(pages))]
    import torch.distributed as dist
    bucket = [None] * ws if rank == 0 else None
    dist.gather_object(local, bucket, dst=0)
    if rank != 0:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(len(pages))]


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
                gather: bool = True, lanes: int = 1):
    """Two-stage alignment of independent (ref, mov) pairs -- mosaic tiles, BASELINE cfg5 -- sharded over the ranks:
    feature-based affine initialisation (FeatureRegistrator, the pipeline's first stage, __main__.py:257-286),
    transform_img_with_tmat, then the optical-flow refinement and warp (OptFlowRegistrator + Warper, :398-433).
    Returns, per pair and in pair order on rank 0, (aligned moving image, 2x3 matrix, flow)."""
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

    return run_sharded(pairs, one, gather=gather, lanes=lanes)
