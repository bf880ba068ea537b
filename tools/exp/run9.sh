set -x
mkdir -p gpurun_out/r05i
python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-variants 2> gpurun_out/r05i/bench_2ranks.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks',d['value'],d['ms_per_step'],d.get('results_to_shared_array_ms'),d.get('results_to_shared_array'))"
python3 tools/soak_stream.py 40 > gpurun_out/r05i/soak_stream.txt 2>&1; tail -2 gpurun_out/r05i/soak_stream.txt
python3 tools/soak_pages.py 1500 > gpurun_out/r05i/soak_pages.txt 2>&1; tail -2 gpurun_out/r05i/soak_pages.txt
python3 tools/soak_parity.py 120 9000 2600 > gpurun_out/r05i/soak_parity_big.txt 2>&1; tail -1 gpurun_out/r05i/soak_parity_big.txt
python -m pytest tests/test_gpu_register.py -x -q -k "page_locked or workspace" 2>&1 | tail -2
