"""Worker for tests/test_gpu_register.py::test_two_ranks_download_into_their_own_page_locked_rows: two ranks (sharing the one
device of the test box), a node-wide shared array whose rows each rank page-locks in ITS mapping (parallel.shared_array,
ws > 1: one registration per owned row, at addresses the .npy header leaves unaligned), downloads into own and foreign rows."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from microaligner_amd import device, parallel  # noqa: E402
from microaligner_amd.device import get_context  # noqa: E402


def main(out_path):
    dist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    ctx = get_context(0)
    rows, shape = 5, (1024, 1536)                     # 6 MB rows: units of work, above the staging threshold
    name = f"ma_gpu_test_{os.environ['MASTER_PORT']}.npy"
    arr = parallel.shared_array(name, (rows,) + shape, np.float32, unlink=True)
    locked = parallel.arr_is_page_locked(arr)
    mine = parallel.shard(rows, rank, ws)
    direct_own = [device.transfer_is_direct(arr[i]) for i in mine]
    direct_other = [device.transfer_is_direct(arr[i]) for i in range(rows) if i not in mine]
    direct_span = device.transfer_is_direct(arr[mine[0]:mine[0] + 2]) if mine[0] + 1 < rows else False
    rng = np.random.default_rng(100 + rank)
    for i in mine:                                     # own rows: by DMA as they are
        src = rng.random(shape).astype(np.float32)
        ctx.asdevice(src).numpy(out=arr[i])
        assert np.array_equal(arr[i], src)
    dist.barrier()
    # every rank sees every row written by its owner
    ok = True
    for r in range(ws):
        g = np.random.default_rng(100 + r)
        for i in parallel.shard(rows, r, ws):
            ok = ok and np.array_equal(arr[i], g.random(shape).astype(np.float32))
    dist.barrier()
    # a download across the end of an owned row (the next row belongs to the other rank: not registered here) is staged
    # and still lands
    for turn in range(ws):                             # one rank at a time: the spans of neighbouring ranks overlap
        if turn == rank and mine[0] + 1 < rows:
            two = rng.random((2,) + shape).astype(np.float32)
            ctx.asdevice(two).numpy(out=arr[mine[0]:mine[0] + 2])
            ok = ok and np.array_equal(arr[mine[0]:mine[0] + 2], two)
        dist.barrier()
    res = [None] * ws
    dist.all_gather_object(res, {"rank": rank, "locked": locked, "direct_own": direct_own, "direct_other": direct_other,
                                 "direct_span": direct_span, "ok": bool(ok)})
    if rank == 0:
        json.dump(res, open(out_path, "w"))
    del arr
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
