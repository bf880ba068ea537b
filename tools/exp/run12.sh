set -x
mkdir -p gpurun_out/r05l
python -m pytest tests/test_feature_reg.py tests/test_gpu_primitives.py -x -q 2>&1 | tail -3
python3 tools/freg_profile.py 4096 > gpurun_out/r05l/freg_profile.txt 2>&1; head -24 gpurun_out/r05l/freg_profile.txt | cut -c1-150
python3 bench.py --workload cfg5 --feature-init --steps 5 --warmup 2 --no-cpu-baseline --no-variants 2>/dev/null | tail -1 | cut -c1-200
