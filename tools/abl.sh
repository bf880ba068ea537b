python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
  python bench.py --workload cfg2 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('cfg2', d['value'], d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/6,2)) for k,v in d['kernels'].items() if k in ('polyexp_m0','blur_v','blur_h_solve')})"
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg3_v9.json
  python -c "
import json; d=json.load(open('gpurun_out/bench_cfg3_v9.json')); print('cfg3', d['value'], d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/3,2)) for k,v in d['kernels'].items()})"
done
