for wlk in cfg4 cfg3; do
  python bench.py --workload $wlk --steps 3 --warmup 1 --no-cpu-baseline > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('$wlk', d['value'], d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/3,2)) for k,v in d['kernels'].items()}, d['roofline_polyexp']['frac'])"
done
