#!/bin/bash
# prologue reorder (first row stages requested before the wait for the resident weights): parity + same-box A/B against the previous build
mkdir -p gpurun_out/r4ad
O=gpurun_out/r4ad
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_ng.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_video.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2; do
for lib in libvlpet_hip_prev.so libvlpet_hip.so; do
  export VLPET_LIB=$PWD/vl-pet_amd/lib/$lib
  K1BENCH_R=96 timeout 300 python tools/k1bench.py $lib 3500 15272 28000 46648 2>&1 | grep k1bench | tee -a $O/k1bench.txt
  K1BENCH_R=192 timeout 300 python tools/k1bench.py $lib 2100 18250 2>&1 | grep k1bench | tee -a $O/k1bench_r192.txt
  K2BENCH_R=96 timeout 300 python tools/k2bench.py $lib 3500 10000 28000 33200 2>&1 | grep k2bench | tee -a $O/k2bench.txt
done; done
