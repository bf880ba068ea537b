for abl in "-DMA_BH_WAVES=1" "-DMA_BH_WAVES=5" "-DMA_BH_WAVES=6" "-DMA_BH_WAVES=5 -DMA_BH_FU=2"; do
  MA_HIPCC_EXTRA="$abl" python -m microaligner_amd.build --force >/dev/null 2>&1
  python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('ABL[$abl]', d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/3,2)) for k,v in d['kernels'].items() if k.startswith('blur')})"
done
