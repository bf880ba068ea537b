#!/bin/bash
# A/B timing of compile-time variants on one box: tools/exp_ab.sh "<flags A>" "<flags B>" ...   (use "" for defaults)
# Each variant is built with MA_HIPCC_EXTRA=<flags> and timed REPS times (interleaved) on cfg3.
cd "$(dirname "$0")/.."
BENCH_ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-variants}
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  MA_HIPCC_EXTRA="$v" python3 -m microaligner_amd.build --force >/dev/null 2>&1 || echo "BUILD FAILED: $v"
  python3 bench.py $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('VARIANT [$v]',d['ms_per_step'],{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k}, 'per launch: blur_h', k['blur_h_solve']['avg_launch_ms'], 'blur_v', k['blur_v']['avg_launch_ms'], 'dog', k.get('dog',{}).get('avg_launch_ms'))
"
done
done
MA_HIPCC_EXTRA="" python3 -m microaligner_amd.build --force >/dev/null 2>&1
