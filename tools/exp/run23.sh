set -x
mkdir -p gpurun_out/r05o
python -m pytest tests -m gpu -q > gpurun_out/r05o/pytest_gpu.log 2>&1; tail -3 gpurun_out/r05o/pytest_gpu.log
python3 tools/soak_parity.py 100 31000 > gpurun_out/r05o/soak.txt 2>&1; tail -1 gpurun_out/r05o/soak.txt
bash tools/collect_profiles.sh r05 > gpurun_out/r05o/collect.log 2>&1; tail -2 gpurun_out/r05o/collect.log
bash tools/kernel_clock.sh r05 > gpurun_out/clock_r05.txt 2>&1; cat gpurun_out/clock_r05.txt
bash tools/sq_counters.sh cfg3 final > gpurun_out/r05o/sq_final.log 2>&1
python3 bench.py > gpurun_out/r05o/bench_default.json 2> gpurun_out/r05o/bench_default.err; tail -c 600 gpurun_out/r05o/bench_default.json
