set -x
mkdir -p gpurun_out/r05c
python -m pytest tests/test_gpu_primitives.py -k "nmi or farneback or gate" -x -q > gpurun_out/r05c/pytest_nmi.log 2>&1; echo "rc=$?" >> gpurun_out/r05c/pytest_nmi.log
tail -3 gpurun_out/r05c/pytest_nmi.log
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants"
one() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$1',d['ms_per_step'],d.get('kernel_time_ms_per_step'),{n:round(k[n]['avg_launch_ms']*k[n]['launches']/d['steps'],2) for n in k})
"; }
for rep in 1 2; do
MICROALIGNER_SIDE_DOG_LDS_KB=0 MICROALIGNER_NMI_BAND_MAX_PX=0 $B 2>/dev/null | one "A old      "
MICROALIGNER_SIDE_DOG_LDS_KB=82 MICROALIGNER_NMI_BAND_MAX_PX=0 $B 2>/dev/null | one "B sideLDS  "
MICROALIGNER_SIDE_DOG_LDS_KB=0 $B 2>/dev/null | one "C band     "
$B 2>/dev/null | one "D both     "
MICROALIGNER_SIDE_DOG_LDS_KB=110 $B 2>/dev/null | one "E both110  "
$B --no-companion 2>/dev/null | one "F nocomp   "
done > gpurun_out/r05c/ab_companion.txt 2>&1
cat gpurun_out/r05c/ab_companion.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r05c/kt_on --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r05c/bench_kt_on.json 2> gpurun_out/r05c/kt_on.err
