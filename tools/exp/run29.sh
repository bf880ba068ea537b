set -x
mkdir -p gpurun_out/r05r
python -m pytest tests -m gpu -q > gpurun_out/r05r/pytest_gpu.log 2>&1; tail -3 gpurun_out/r05r/pytest_gpu.log
python3 tools/soak_parity.py 120 41000 > gpurun_out/r05r/soak.txt 2>&1; tail -1 gpurun_out/r05r/soak.txt
bash tools/collect_profiles.sh r05 > gpurun_out/r05r/collect.log 2>&1; tail -2 gpurun_out/r05r/collect.log
bash tools/kernel_clock.sh r05 > gpurun_out/clock_r05.txt 2>&1; cat gpurun_out/clock_r05.txt
bash tools/sq_counters.sh cfg3 final > gpurun_out/r05r/sq_final.log 2>&1
python3 bench.py > gpurun_out/r05r/bench_default.json 2> gpurun_out/r05r/bench_default.err; tail -c 600 gpurun_out/r05r/bench_default.json
