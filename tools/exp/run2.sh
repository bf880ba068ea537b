set -x
mkdir -p gpurun_out/r05b
python -m pytest tests/test_gpu_primitives.py -k "farneback" -x -q > gpurun_out/r05b/pytest_fb.log 2>&1; echo "rc=$?" >> gpurun_out/r05b/pytest_fb.log
tail -3 gpurun_out/r05b/pytest_fb.log
bash tools/ab_libs.sh 2 $PWD/variants/libma_bv64.so tree > gpurun_out/r05b/ab_bv32.txt 2>&1
cat gpurun_out/r05b/ab_bv32.txt
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_register.py -x -q > gpurun_out/r05b/pytest_full.log 2>&1; echo "rc=$?" >> gpurun_out/r05b/pytest_full.log
tail -3 gpurun_out/r05b/pytest_full.log
