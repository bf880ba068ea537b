set -x
mkdir -p gpurun_out/r05a
python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05a/pytest_gpu.log
MICROALIGNER_HIP_LIB=$PWD/variants/libma_rows32.so python -m pytest tests/test_gpu_primitives.py -k "farneback" -x -q > gpurun_out/r05a/pytest_rows32.log 2>&1; echo "rc=$?" >> gpurun_out/r05a/pytest_rows32.log
bash tools/ab_libs.sh 2 $PWD/variants/libma_base.so $PWD/variants/libma_rows32.so > gpurun_out/r05a/ab_rows32.txt 2>&1
bash tools/sq_counters.sh cfg3 base > gpurun_out/r05a/sq_base.log 2>&1
MICROALIGNER_HIP_LIB=$PWD/variants/libma_rows32.so bash tools/sq_counters.sh cfg3 rows32 > gpurun_out/r05a/sq_rows32.log 2>&1
