"""Worker for tests/test_multi_process.py: world-size-2 gloo run of the sharding driver on CPU.
The compute function is the CPU oracle (allowed in tests): what is under test is the partition, the gather
and the ordering of microaligner_amd.parallel, i.e. the N > 1 path of bench.py / the pipeline."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from microaligner_amd import parallel, synthetic  # noqa: E402
from oracle import register_oracle as RO  # noqa: E402


def main(out_path):
    dist.init_process_group("gloo")
    rank, ws = dist.get_rank(), dist.get_world_size()
    assert parallel.world() == (rank, ws)
    params = dict(num_pyr_lvl=1, use_full_res_img=True, tile_size=100, overlap=12)
    pairs = [synthetic.make_pair(210, 220, seed) for seed in range(1, 6)]  # 5 units over 2 ranks: 3 + 2

    def compute(pair):
        flow, _ = RO.register(pair[0], pair[1], **params)
        return flow

    calls = []

    def counted(pair):
        calls.append(1)
        return compute(pair)

    gathered = parallel.run_sharded(pairs, counted, gather=True)
    local = parallel.run_sharded(pairs, compute, gather=False)
    n_calls = len(calls)
    assert sorted(local) == parallel.shard(len(pairs), rank, ws)
    assert n_calls == len(parallel.shard(len(pairs), rank, ws))

    # results in place: every rank writes its flows into a node-wide shared array; units are loaders, evaluated by their
    # owner only; nothing is gathered
    loaded = []

    def loader(k):
        def load():
            loaded.append(k)
            return pairs[k]
        return load

    store = parallel.shared_array(f"ma_test_{os.environ['MASTER_PORT']}", (len(pairs), 210, 220, 2), np.float32)
    got = parallel.run_sharded([loader(k) for k in range(len(pairs))], compute, out=store)
    assert got is store and sorted(loaded) == parallel.shard(len(pairs), rank, ws)
    shared_ok = all(np.array_equal(store[k], compute(pairs[k])) for k in range(len(pairs)))   # every rank sees every row
    parallel.shared_array_unlink(f"ma_test_{os.environ['MASTER_PORT']}")
    assert shared_ok

    # timing reduction used by bench.py: max over ranks
    import torch
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(ws)

    if rank == 0:
        assert gathered is not None and len(gathered) == len(pairs)
        ok = all(np.array_equal(g, compute(p)) for g, p in zip(gathered, pairs))
        json.dump({"ok": bool(ok), "world": ws, "units": len(pairs)}, open(out_path, "w"))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
