"""The N > 1 path on CPU: two processes, gloo, 127.0.0.1 (no GPU needed)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from microaligner_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_is_a_partition():
    for n in (0, 1, 5, 8, 17):
        for ws in (1, 2, 3, 8):
            parts = [parallel.shard(n, r, ws) for r in range(ws)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        parallel.shard(4, 2, 2)


def test_single_process_run_sharded():
    assert parallel.run_sharded([1, 2, 3], lambda v: v * v) == [1, 4, 9]
    assert parallel.run_sharded([1, 2, 3], lambda v: v * v, gather=False) == {0: 1, 1: 4, 2: 9}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gloo(tmp_path):
    out = tmp_path / "result.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_dist_worker.py"), str(out)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.load(open(out))
    assert res == {"ok": True, "world": 2, "units": 5}


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must itself start two ranks (the launcher
    never touches HIP), rendezvous them over gloo and report n_gpus: 2.  --dry-run skips the GPU work so that the
    launch / barrier / max-reduce / JSON plumbing is testable here."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["config"]["pairs_per_step"] == 2
    assert set(res["host_modes"]) >= {"stream_pairs_pageable", "stream_pairs_page_locked", "warp_pages_pageable",
                                      "warp_pages_page_locked"}


def test_bench_eight_ranks_report_per_rank_rows_and_deal_pairs_round_robin():
    """What the driver's 8-GPU run exercises, without GPUs: eight ranks rendezvous, every rank contributes a row (its
    time per step, device identity, the pairs it owns), rank 0 prints min / mean / max over the ranks; with --pairs-total
    the 64 mosaic tiles of BASELINE cfg5 are dealt round-robin (strong scaling)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "0",
                        "--dry-run", "--workload", "cfg5", "--pairs-total", "64"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 8 and res["config"]["pairs_per_step"] == 64
    assert [row["rank"] for row in res["ranks"]] == list(range(8))
    assert [row["pairs"] for row in res["ranks"]] == [list(range(r0, 64, 8)) for r0 in range(8)]
    # the data plane: every unit was loaded by the rank that owns it and by nobody else, and its result reached the
    # node-wide shared array without passing through the control plane
    assert [row["units_loaded"] for row in res["ranks"]] == [list(range(r0, 64, 8)) for r0 in range(8)]
    assert res["shared_results_ok"] is True and "shared memory" in res["results_via"]
    per = [row["ms_per_step"] for row in res["ranks"]]
    assert res["rank_ms_per_step"] == {"min": min(per), "mean": sum(per) / 8, "max": max(per)}
    assert abs(res["ms_per_step"] - max(per)) < 1e-9            # the headline time is the slowest rank's
    # the numpy -> numpy modes (the ones that can fail to scale) are part of every N > 1 line: per mode one
    # time per rank, the whole-node rates from the slowest rank, and whether the library moved the buffers by DMA as they are
    hm = res["host_modes"]
    for mode in ("stream_pairs_pageable", "stream_pairs_page_locked", "warp_pages_pageable", "warp_pages_page_locked"):
        row = hm[mode]
        assert len(row["ms_per_unit_per_rank"]) == 8 and len(row["buffers_moved_directly"]) == 8
        assert row["aggregate_mpix_s"] > 0 and row["units_per_rank"] == 8
        assert row["buffers_moved_directly"] == [mode.endswith("locked")] * 8
        slowest = max(row["ms_per_unit_per_rank"]) * row["units_per_rank"] / 1e3
        assert row["aggregate_mpix_s"] == pytest.approx(8 * row["units_per_rank"] * 1e6 / slowest / 1e6, rel=1e-2)
    assert len(hm["host_register_ms_per_gib"]) == 8


def test_bench_launcher_propagates_a_rank_failure():
    """Without a HIP device the ranks fail; the launcher must stop the survivors and exit non-zero, not hang."""
    from microaligner_amd import device
    try:
        if device.device_count() > 0:
            pytest.skip("a HIP device is present: the ranks would run the real benchmark")
    except Exception:
        pass
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--workload", "cfg1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_eight_rank_host_soak_runs_without_a_device():
    """tools/host_soak.py: eight processes, each bound to its share of the host's NUMA nodes, cycle the
    staging copies of the numpy -> numpy stream in both directions at once through the library's two copy pools -- no device
    involved.  Here: a miniature (the numbers of the 2 x 64-core GPU host are in DESIGN.md); every rank must have moved bytes
    in both directions and the report must carry the needed-vs-sustained fields."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_soak.py"), "--ranks", "8", "--scale", "0.004",
                        "--seconds", "0.5", "--mode", "stream"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.strip().splitlines()]
    assert "numa_nodes" in lines[0]
    row = lines[1]
    assert row["ranks"] == 8 and len(row["per_rank_in_gb_s"]) == 8 and len(row["per_rank_out_gb_s"]) == 8
    assert min(row["per_rank_in_gb_s"]) > 0 and min(row["per_rank_out_gb_s"]) > 0
    assert row["needed_in_gb_s"] == pytest.approx(8 * 2 * 16384 ** 2 * 4 / 0.075 / 1e9, rel=1e-3)
    assert 0 < row["covers"]


def test_bench_host_memory_guard_for_the_host_modes():
    """The numpy -> numpy modes hold ~10 GiB of host arrays per rank; bench.py sizes them against what the node can still
    give (MemAvailable, the container's limit) before every rank allocates -- a rank killed for memory takes the headline."""
    sys.path.insert(0, ROOT)
    import bench
    room, how = bench.available_host_bytes()
    assert room is None or (room > 0 and how in ("MemAvailable", "cgroup memory limit"))
    one = bench.host_modes_bytes(16384, 16384, 4, 8)
    assert 10 * 2 ** 30 < one < 16 * 2 ** 30                       # cfg3: 4 + 6 GiB of stream arrays, staging, slack
    assert bench.host_modes_bytes(1000, 1000, 4, 8) < 4 * 2 ** 30    # the slack dominates small workloads
    assert bench.host_modes_bytes(16384, 16384, 1, 32) > one          # many pages outgrow the stream


def test_bench_line_survives_a_rank_that_fails_inside_the_host_modes():
    """The headline is gathered before the numpy -> numpy modes start, and those synchronise over a store of their own with
    bounded waits: a rank that fails half way (here: rank 1 before the third mode's barrier) costs the others the timeout and
    shows up as an error row -- the line is printed and the job ends.  (A 4-rank rehearsal on one GPU lost its line to a
    30-minute gloo timeout before this: a torch.distributed collective entered by some ranks only.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    env["MA_BENCH_DRY_FAIL"] = "1:2"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "0",
                        "--dry-run", "--host-mode-timeout", "4"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    # (no wall-clock bound here: the first `import torch` of three ranks on a cold box takes minutes by itself; a rank that waited
    # for a collective instead of the 4-second barrier would fail the content checks below -- or the process group's timeout)
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["n_gpus"] == 3 and [row["rank"] for row in res["ranks"]] == [0, 1, 2]       # the headline's rows are whole
    hm = res["host_modes"]
    for mode in ("stream_pairs_pageable", "stream_pairs_page_locked"):                        # measured by all three before the failure
        assert len(hm[mode]["ms_per_unit_per_rank"]) == 3 and "error" not in hm[mode]
    # the third mode: rank 1 never reached its barrier; the others waited out the timeout and left the modes too
    assert "warp_pages_pageable" in hm and hm["warp_pages_pageable"].get("error")
    assert "injected failure" in json.dumps(hm["errors"]["host_modes_error"]["rank 1"])
    assert "barrier 3 not reached" in json.dumps(hm["errors"]["host_modes_error"])
