#!/bin/bash
# A/B timing of two BUILDS of the library on one box (interleaved): tools/exp_lib_ab.sh <base .so> [reps]
# The tree's own build is B; MICROALIGNER_HIP_LIB selects A.  Prints ms per step and per kernel group.
cd "$(dirname "$0")/.."
BASE=$1; REPS=${2:-2}
BENCH_ARGS=${BENCH_ARGS:---steps 6 --warmup 2 --no-cpu-baseline --no-variants}
for rep in $(seq 1 $REPS); do
for v in "$BASE" ""; do
  MICROALIGNER_HIP_LIB="$v" python3 bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('BUILD [${v:-tree}]',d['ms_per_step'],d['kernel_time_ms_per_step'],{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"
done
done
