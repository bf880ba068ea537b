for rep in 1 2 3; do
for abl in "-DMA_BH_WAVES=4" "-DMA_BH_WAVES=6"; do
  MA_HIPCC_EXTRA="$abl" python -m microaligner_amd.build --force >/dev/null 2>&1
  python bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('ABL[$abl]', d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/6,2)) for k,v in d['kernels'].items() if k.startswith('blur')})"
done
done
