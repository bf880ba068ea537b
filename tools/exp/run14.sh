set -x
mkdir -p gpurun_out/r05m
python -m pytest tests -m gpu -x -q > gpurun_out/r05m/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05m/pytest_gpu.log
tail -3 gpurun_out/r05m/pytest_gpu.log
bash tools/collect_profiles.sh r05 > gpurun_out/r05m/collect.log 2>&1; tail -3 gpurun_out/r05m/collect.log
bash tools/kernel_clock.sh r05 > gpurun_out/clock_r05.txt 2>&1; cat gpurun_out/clock_r05.txt
bash tools/sq_counters.sh cfg3 final > gpurun_out/r05m/sq_final.log 2>&1; tail -3 gpurun_out/r05m/sq_final.log
python3 bench.py > gpurun_out/r05m/bench_default.json 2> gpurun_out/r05m/bench_default.err; tail -c 1500 gpurun_out/r05m/bench_default.json
