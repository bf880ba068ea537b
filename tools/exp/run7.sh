set -x
mkdir -p gpurun_out/r05g
bash tools/ab_libs.sh 2 $PWD/variants/libma_base2.so $PWD/variants/libma_prio1.so $PWD/variants/libma_prio2.so > gpurun_out/r05g/ab_prio.txt 2>&1
cat gpurun_out/r05g/ab_prio.txt
