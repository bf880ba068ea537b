#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the page-warp driver (tools/page_rate.py): bash tools/profile_page_warp.sh r04
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_pages_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/t --output-format csv -- python3 tools/page_rate.py > $OUT/page_rate.txt 2> $OUT/err
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, re, sys
out, tag = sys.argv[1], sys.argv[2]
rows, cps = [], []
for f in glob.glob(out + "/t/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"::(\w+)", r["Name"])
        rows.append((m.group(1) if m else r["Name"][:40], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
for f in glob.glob(out + "/t/*/*_memory_copy_stats.csv"):
    for r in csv.DictReader(open(f)):
        cps.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
with open(f"{out}/{tag}_page_warp_trace.csv", "w") as fh:
    fh.write("kind,name,calls,total_ms,avg_us\n")
    for name, calls, tot, avg in sorted(rows, key=lambda r: -r[2]):
        fh.write(f"kernel,{name},{calls},{tot:.3f},{avg:.2f}\n")
    for name, calls, tot, avg in cps:
        fh.write(f"copy,{name},{calls},{tot:.3f},{avg:.2f}\n")
print(open(f"{out}/{tag}_page_warp_trace.csv").read())
print(open(out + "/page_rate.txt").read())
PY
