set -x
mkdir -p gpurun_out/r05f
python3 tools/soak_parity.py 160 5000 > gpurun_out/r05f/soak_parity.txt 2>&1; tail -3 gpurun_out/r05f/soak_parity.txt
bash tools/collect_profiles.sh r05 > gpurun_out/r05f/collect.log 2>&1; tail -5 gpurun_out/r05f/collect.log
bash tools/kernel_clock.sh r05 > gpurun_out/clock_r05.txt 2>&1; cat gpurun_out/clock_r05.txt
python3 tools/freg_profile.py 4096 > gpurun_out/r05f/freg_profile.txt 2>&1; head -3 gpurun_out/r05f/freg_profile.txt
python3 bench.py --workload cfg5 --feature-init --steps 5 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 | cut -c1-300
MICROALIGNER_COPY_THREADS=4 python3 tools/host_soak.py --ranks 1,2 --mode stream --seconds 3 > gpurun_out/r05f/host_soak_4threads.txt 2>&1; cat gpurun_out/r05f/host_soak_4threads.txt | cut -c1-400
python3 bench.py > gpurun_out/r05f/bench_default.json 2> gpurun_out/r05f/bench_default.err; tail -c 3000 gpurun_out/r05f/bench_default.json
