#!/bin/bash
mkdir -p gpurun_out/r4l
O=gpurun_out/r4l
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -4 | tee $O/log.txt
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so K1BENCH_R=96
for rep in 1 2; do
for v in 1 2 4; do
VLPET_DZ2_FSPLIT=$v timeout 300 python tools/k1bench.py fsplit=$v 1000 2100 3500 5000 8192 12000 15272 2>&1 | grep k1bench | tee -a $O/k1bench_small.txt
done; done
