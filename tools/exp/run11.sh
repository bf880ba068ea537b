set -x
mkdir -p gpurun_out/r05k
python -m pytest tests/test_gpu_register.py tests/test_gpu_primitives.py -x -q 2>&1 | tail -2
python3 tools/soak_parity.py 140 15000 > gpurun_out/r05k/soak_parity.txt 2>&1; tail -1 gpurun_out/r05k/soak_parity.txt
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$1',d['ms_per_step'],d.get('kernel_time_ms_per_step'),{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"; }
for rep in 1 2 3; do
MICROALIGNER_HIP_LIB=$PWD/variants/libma_head2.so $B 2>/dev/null | one "head "
$B 2>/dev/null | one "tree "
MICROALIGNER_HIP_LIB=$PWD/variants/libma_head2.so $B --no-companion 2>/dev/null | one "head-nocomp "
$B --no-companion 2>/dev/null | one "tree-nocomp "
done > gpurun_out/r05k/ab.txt 2>&1
grep -E "^(head|tree)" gpurun_out/r05k/ab.txt
python -m pytest tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
