#!/bin/bash
# Collects the rocprofv3 evidence for profiles/: kernel trace + stats, and HBM traffic counters in separate
# --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box:
#   bash tools/collect_profiles.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
# Per-kernel numbers are collected with the companion stream OFF (--no-companion: MA_OPT_COMPANION_STREAM = 0): every
# launch alone on the chip, so that averages are comparable from run to run and round to round.  The step time is what the
# product does (companion ON): recorded by the un-profiled runs at the end.
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants --no-companion"
rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- $CMD > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
rocprofv3 --kernel-trace --stats -d $OUT/kt_on --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > $OUT/bench_under_kernel_trace_companion_on.json 2> $OUT/kt_on.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch --output-format csv -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/write --output-format csv -- $CMD > /dev/null 2> $OUT/write.err
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $OUT/bench_companion_on.json 2> /dev/null
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-companion > $OUT/bench_companion_off.json 2> /dev/null
python3 tools/summarise_profiles.py $OUT $TAG
# gpurun only brings gpurun_out/ back: leave copies of the summaries there (commit them under profiles/ from the build container)
mkdir -p $OUT/summary && cp profiles/${TAG}_kernel_stats_cfg3.csv profiles/${TAG}_kernel_stats_cfg3_companion_on.csv profiles/${TAG}_hbm_traffic_cfg3.json $OUT/summary/
cp $OUT/bench_companion_on.json $OUT/bench_companion_off.json $OUT/summary/
