#!/bin/bash
# A/B timing of SOURCE variants of one HIP file on one box (interleaved, two repetitions):
#   tools/exp_src_ab.sh farneback.hip tools/exp/farneback_a.hip tools/exp/farneback_b.hip ...
# "-" stands for the file as it is in the tree.  Prints ms per step and ms per step and kernel group.
cd "$(dirname "$0")/.."
TARGET=microaligner_amd/csrc/$1; shift
BENCH_ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-variants}
cp $TARGET /tmp/_ab_orig.hip
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "-" ]; then cp /tmp/_ab_orig.hip $TARGET; else cp "$v" $TARGET; fi
  python3 -m microaligner_amd.build >/dev/null 2>&1 || echo "BUILD FAILED: $v"
  python3 bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('VARIANT [$v]',d['ms_per_step'],{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k}, 'blur_h/launch', k['blur_h_solve']['avg_launch_ms'], 'blur_v/launch', k['blur_v']['avg_launch_ms'])
"
done
done
cp /tmp/_ab_orig.hip $TARGET
python3 -m microaligner_amd.build >/dev/null 2>&1
