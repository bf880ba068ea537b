set -x
mkdir -p gpurun_out/r05d
python -m pytest tests -m gpu -x -q > gpurun_out/r05d/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05d/pytest_gpu.log
tail -3 gpurun_out/r05d/pytest_gpu.log
python3 tools/host_soak.py --ranks 1,2,4,8 --mode stream --seconds 3 > gpurun_out/r05d/host_soak_stream.txt 2>&1
python3 tools/host_soak.py --ranks 1,2,4,8 --mode pages --seconds 3 > gpurun_out/r05d/host_soak_pages.txt 2>&1
cat gpurun_out/r05d/host_soak_stream.txt gpurun_out/r05d/host_soak_pages.txt
python3 tools/freg_profile.py 4096 > gpurun_out/r05d/freg_profile.txt 2>&1
head -50 gpurun_out/r05d/freg_profile.txt
python3 bench.py --workload cfg5 --feature-init --steps 5 --warmup 2 --no-cpu-baseline --no-variants > gpurun_out/r05d/bench_cfg5_freg.json 2>/dev/null
