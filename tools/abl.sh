python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/b.json
python -c "
import json; d=json.load(open('/tmp/b.json')); print('cfg3', d['value'], d['ms_per_step'], {k:(round(v['avg_launch_ms']*v['launches']/3,2)) for k,v in d['kernels'].items()}, d['roofline_polyexp']['frac'])"
done
