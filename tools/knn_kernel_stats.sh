#!/bin/bash
# rocprofv3 kernel statistics of tools/exp_knn.py (n x n descriptors of 200 floats, exact and filtered search twice each).
# Run on the GPU box: bash tools/knn_kernel_stats.sh [n]; the table lands in gpurun_out/knn_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/prof_knn
rm -rf $OUT; mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format csv -- python3 tools/exp_knn.py ${1:-45000} > $OUT/out.txt 2> $OUT/err.txt
F=$(find $OUT/kt -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then cp "$F" gpurun_out/knn_kernel_stats.csv; head -8 "$F" | cut -c1-150; else echo "no kernel stats"; tail -5 $OUT/err.txt; fi
