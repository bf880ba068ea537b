#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / dispatch duration, dispatches of
# >= 0.3 ms only (MI355X_MICROARCH.md, DVFS section), companion stream off (every kernel alone on the chip).
# bash tools/kernel_clock.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-clk}; shift
OUT=gpurun_out/clock_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-variants --no-companion "$@" > /dev/null 2> $OUT/err
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(sys.argv[1] + "/p/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if dt < 3e5:
            continue
        m = re.search(r"::(\w+)", r["Kernel_Name"])
        a = acc[m.group(1) if m else r["Kernel_Name"][:40]]
        a[0] += 1; a[1] += float(r["Counter_Value"]) / 8; a[2] += dt
for k, (n, cyc, ns) in sorted(acc.items()):
    print(f"{k:28s} launches {n:4d}  avg {ns / n / 1e6:7.3f} ms  clock {cyc / ns:5.3f} GHz")
PY
