set -x
mkdir -p gpurun_out/r05q
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$1',d['ms_per_step'],d.get('kernel_time_ms_per_step'),d.get('sustained_clock_ghz'),{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"; }
for rep in 1 2 3; do
MICROALIGNER_NMI_BAND_MAX_PX=0 MICROALIGNER_HIP_LIB=$PWD/variants/libma_r04k.so $B 2>/dev/null | one "round4-kernels "
$B 2>/dev/null | one "round5-kernels "
MICROALIGNER_HIP_LIB=$PWD/variants/libma_dogo1.so $B 2>/dev/null | one "round5+dog-plain "
MICROALIGNER_NMI_BAND_MAX_PX=0 MICROALIGNER_HIP_LIB=$PWD/variants/libma_r04k.so $B --no-companion 2>/dev/null | one "round4-kernels-nocomp "
$B --no-companion 2>/dev/null | one "round5-kernels-nocomp "
MICROALIGNER_HIP_LIB=$PWD/variants/libma_dogo1.so $B --no-companion 2>/dev/null | one "round5+dog-plain-nocomp "
done > gpurun_out/r05q/ab_r04_r05.txt 2>&1
grep -E "^round" gpurun_out/r05q/ab_r04_r05.txt
