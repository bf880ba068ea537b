#!/bin/bash
# rocprofv3 kernel statistics of FeatureRegistrator.register() on one 4096^2 mosaic tile (tools/freg_profile.py).
# Run on the GPU box: bash tools/freg_kernel_stats.sh ; the table lands in gpurun_out/freg_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_freg
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python3 tools/freg_profile.py 4096 > $OUT/out.txt 2> $OUT/err.txt
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" gpurun_out/freg_kernel_stats.csv; head -25 "$F" | cut -c1-160; else echo "no kernel stats"; tail -5 $OUT/err.txt; fi
