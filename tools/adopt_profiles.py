#!/usr/bin/env python3
"""Copies the profile summaries a gpurun call left under gpurun_out/ into profiles/ (tracked), refusing summaries that
were not collected with the kernels of this tree:

    python3 tools/adopt_profiles.py r03 [clock.txt] [sq_counters.txt]

gpurun_out/prof_<tag>/summary/ comes from tools/collect_profiles.sh, clock.txt from tools/kernel_clock.sh, the SQ
counter table from tools/sq_counters.sh."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from microaligner_amd import build  # noqa: E402

tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "summary")
traffic = json.load(open(os.path.join(src, f"{tag}_hbm_traffic_cfg3.json")))
want = build.source_hash()
if traffic.get("kernel_source_hash") != want:
    sys.exit(f"the summaries were collected with kernel sources {traffic.get('kernel_source_hash')}, this tree is {want}")
for name in (f"{tag}_kernel_stats_cfg3.csv", f"{tag}_kernel_stats_cfg3_companion_on.csv", f"{tag}_hbm_traffic_cfg3.json"):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(ROOT, "profiles", name))
# the GPU box has no .git: the commit the measured tree descends from is recorded here, at adoption
import subprocess
head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
traffic["head_when_adopted"] = head
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic_cfg3.json"), "w"), indent=1)
if len(sys.argv) > 2:
    clk = {}
    for ln in open(sys.argv[2]):
        m = re.match(r"(\S+)\s+launches\s+(\d+)\s+avg\s+([\d.]+) ms\s+clock\s+([\d.]+) GHz", ln)
        if m:
            clk[m.group(1)] = {"launches": int(m.group(2)), "avg_ms_profiled": float(m.group(3)), "clock_ghz": float(m.group(4))}
    json.dump({"kernel_source_hash": want,
               "command": "rocprofv3 --pmc GRBM_GUI_ACTIVE -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "
                          "--no-variants (cfg3), tools/kernel_clock.sh",
               "note": "effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, dispatches >= 0.3 ms "
                       "(MI355X_MICROARCH.md, DVFS); profiled passes run a few percent below unprofiled ones",
               "per_kernel": clk}, open(os.path.join(ROOT, "profiles", f"{tag}_kernel_clocks_cfg3.json"), "w"), indent=1)
if len(sys.argv) > 3:
    shutil.copy(sys.argv[3], os.path.join(ROOT, "profiles", f"{tag}_sq_counters_cfg3.txt"))
print("adopted", tag, "for kernel sources", want)
