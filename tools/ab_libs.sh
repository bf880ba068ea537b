#!/bin/bash
# A/B timing of library builds on one box, interleaved: tools/ab_libs.sh <reps> <lib or "tree"> ...
# BENCH_ARGS overrides the bench command line.  Prints ms per step, per kernel group and per launch of the two blurs.
cd "$(dirname "$0")/.."
REPS=$1; shift
BENCH_ARGS=${BENCH_ARGS:---steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-companion}
for rep in $(seq 1 $REPS); do
for v in "$@"; do
  lib="$v"; [ "$v" = "tree" ] && lib=""
  MICROALIGNER_HIP_LIB="$lib" python3 bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('BUILD [$v]',d['ms_per_step'],d.get('kernel_time_ms_per_step'),{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"
done
done
