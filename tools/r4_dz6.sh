#!/bin/bash
mkdir -p gpurun_out/r4j
O=gpurun_out/r4j
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 | tee $O/log.txt
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so K1BENCH_R=192
for rep in 1 2; do
for v in 0 1; do
VLPET_DZ6=$v timeout 300 python tools/k1bench.py dz6=$v 2100 3500 9200 16800 18250 28000 2>&1 | grep k1bench | tee -a $O/k1bench_r192.txt
done; done
