#!/bin/bash
# Samples rocm-smi clocks / power while the cfg3 bench loop runs (read-only queries).
cd "$(dirname "$0")/.."
python3 bench.py --steps 60 --warmup 2 --no-cpu-baseline --no-variants > /tmp/clock_probe_bench.json 2>/dev/null &
BP=$!
sleep 8
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | tr -s ' ' | head -6
  echo ---
  sleep 0.7
done
wait $BP
python3 -c "
import json; d=json.loads(open('/tmp/clock_probe_bench.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
