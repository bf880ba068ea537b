"""Turns the rocprofv3 CSVs of tools/collect_profiles.sh into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
os.makedirs(prof, exist_ok=True)


def short(name):
    m = re.search(r"::(\w+)", name)
    return m.group(1) if m else name.split("(")[0][:40]


GROUP = {"fb_polyexp_m0": "polyexp_m0", "fb_blur_v": "blur_v", "fb_blur_v_stream": "blur_v", "fb_blur_h_solve": "blur_h_solve",
         "warp_tiled_kernel": "warp", "window_max_kernel": "merge", "cell_max_kernel": "merge",
         "window_from_cells_kernel": "merge", "merge_flows_kernel": "merge",
         "pyr_down_kernel": "pyr_down", "pyr_up_flow_kernel": "pyr_up", "dog_rows": "dog", "dog_cols_diff": "dog",
         "dog_fused": "dog",
         "scale_to_u8": "dog", "minmax_partial": "dog", "minmax_final": "dog", "dog_params_in": "dog",
         "dog_params_out": "dog", "joint_hist_kernel": "nmi", "joint_hist16_kernel": "nmi", "nmi_reduce_kernel": "nmi"}

def newest(pattern):
    """gpurun merges new files next to those of earlier calls: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)


# 1. kernel stats (rocprofv3 --kernel-trace --stats): companion stream off (every kernel alone on the chip: the numbers
#    that are comparable from round to round) and, when collected, the default command (companion on: what bench.py's own
#    per-kernel events see in the default run)
def write_stats(sub, name):
    found = glob.glob(os.path.join(out, sub, "*", "*_kernel_stats.csv"))
    if not found:
        return
    rows = list(csv.DictReader(open(max(found, key=os.path.getmtime))))
    with open(os.path.join(prof, name), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}",
                        f"{float(r['AverageNs']) / 1e3:.2f}", r["Percentage"]])


write_stats("kt", f"{tag}_kernel_stats_cfg3.csv")
write_stats("kt_on", f"{tag}_kernel_stats_cfg3_companion_on.csv")

# 2. HBM traffic per kernel (separate --pmc passes; FETCH_SIZE is doubled per MI355X_MICROARCH.md, HBM section)
tot = {}
for name in ("fetch", "write"):
    f = newest(os.path.join(out, name, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    tot[name] = agg
traffic = {}
for k, (n, kb) in tot["fetch"].items():
    wkb = tot["write"].get(k, [0, 0.0])[1]
    traffic[k] = {"launches": n, "fetch_bytes_raw": kb * 1024, "fetch_bytes_x2": kb * 2048, "write_bytes": wkb * 1024,
                  "hbm_bytes_per_launch": (kb * 2048 + wkb * 1024) / max(n, 1)}
groups = collections.defaultdict(lambda: {"hbm_bytes": 0.0})
for k, v in traffic.items():
    g = GROUP.get(k)
    if g:
        groups[g]["hbm_bytes"] += v["fetch_bytes_x2"] + v["write_bytes"]
steps = 3  # --steps 2 --warmup 1
sys.path.insert(0, root)
from microaligner_amd import _lib  # noqa: E402
json.dump({"kernel_source_hash": _lib.source_hash(), "library": _lib.load().ma_version().decode(),
           "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-companion (cfg3)",
           "steps_profiled": steps,
           "note": "FETCH_SIZE*2 + WRITE_SIZE (KiB -> bytes), per MI355X_MICROARCH.md HBM section; separate --pmc passes",
           "per_kernel": traffic,
           "per_bench_group_bytes_per_step": {g: v["hbm_bytes"] / steps for g, v in groups.items()}},
          open(os.path.join(prof, f"{tag}_hbm_traffic_cfg3.json"), "w"), indent=1, sort_keys=True)
print("wrote", os.listdir(prof))
