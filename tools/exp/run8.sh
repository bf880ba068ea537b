set -x
mkdir -p gpurun_out/r05h
python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants > gpurun_out/r05h/bench_2ranks.json 2> gpurun_out/r05h/bench_2ranks.err; tail -c 1500 gpurun_out/r05h/bench_2ranks.json; tail -5 gpurun_out/r05h/bench_2ranks.err
python3 bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2',d['ms_per_step'],d['value'])"
python3 bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4',d['ms_per_step'],d['value'])"
python3 bench.py --workload cfg1 --steps 20 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg1',d['ms_per_step'],d['value'])"
python3 bench.py --workload cfg5 --steps 10 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5',d['ms_per_step'],d['value'])"
python3 -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')
"
