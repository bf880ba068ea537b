#!/bin/bash
# A/B timing of compile-time variants on one box: tools/exp_ab.sh "<flags A>" "<flags B>" ...   (use "" for defaults)
# Each variant is built with MA_HIPCC_EXTRA=<flags> and timed twice (interleaved) on cfg3 without DOG.
cd "$(dirname "$0")/.."
BENCH_ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-dog}
for rep in 1 2; do
for v in "$@"; do
  MA_HIPCC_EXTRA="$v" python3 -m microaligner_amd.build --force >/dev/null 2>&1 || echo "BUILD FAILED: $v"
  python3 bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('VARIANT [$v]',d['ms_per_step'],{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"
done
done
MA_HIPCC_EXTRA="" python3 -m microaligner_amd.build --force >/dev/null 2>&1
