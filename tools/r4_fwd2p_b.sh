#!/bin/bash
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
export HIP_FORCE_DEV_KERNARG=1
echo "== K1 gpu tests" | tee $O/log.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cols.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_video.py tests/test_host_golden.py -m gpu -q 2>&1 | tail -15 | tee -a $O/log.txt
echo "== forward A/B (debug build)" | tee -a $O/log.txt
export K1BENCH_FWD_ONLY=1
for mode in 0 1 2 3; do
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_FWD2P=$mode K1BENCH_R=96 timeout 300 python tools/k1bench.py fwd2p=$mode 2100 3500 15272 28000 31616 33200 46648 2>&1 | grep k1bench | tee -a $O/k1fwd_r96.txt
done
for mode in 0 1 2 3; do
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_FWD2P=$mode K1BENCH_R=192 timeout 300 python tools/k1bench.py fwd2p=$mode 2100 3500 9200 16800 18250 28000 2>&1 | grep k1bench | tee -a $O/k1fwd_r192.txt
done
echo "== stamps" | tee -a $O/log.txt
for mode in 2 3; do
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_f2s.so VLPET_FWD2P=$mode K1BENCH_R=96 K1BENCH_ITERS=1 timeout 120 python tools/k1bench.py stamps=$mode 28000 2>&1 | grep -E "f2 stamps" | tail -12 | tee -a $O/stamps_r96.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_f2s.so VLPET_FWD2P=$mode K1BENCH_R=192 timeout 120 python tools/k1bench.py stamps=$mode 18250 2>&1 | grep -E "f2 stamps" | tail -12 | tee -a $O/stamps_r192.txt
done
