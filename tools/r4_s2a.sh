#!/bin/bash
# round 4, second session, first GPU run: parity of the two-pass K2 / K3 forward + the one-trip finalize, then same-box A/Bs (debug build)
mkdir -p gpurun_out/r4m
O=gpurun_out/r4m
export HIP_FORCE_DEV_KERNARG=1
echo "== K2 / K3 / K1 gpu tests" | tee $O/log.txt
timeout 1200 python -m pytest tests/test_gpu_ng.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_cols.py tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -15 | tee -a $O/log.txt
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for mode in 0 -1; do
  for M in 2500 10000 28000 46648; do
    VLPET_FWD2P=$mode timeout 300 python tools/k3bench.py $M 2>&1 | grep "r=" | sed "s/^/k3bench fwd2p=$mode M=$M /" | tee -a $O/k3bench.txt
  done
  VLPET_FWD2P=$mode K2BENCH_R=96 timeout 300 python tools/k2bench.py fwd2p=$mode 2100 3500 10000 15272 28000 31616 33200 46648 2>&1 | grep k2bench | tee -a $O/k2bench.txt
done
K1BENCH_R=96 timeout 300 python tools/k1bench.py fin48 3500 15272 28000 46648 2>&1 | grep k1bench | tee -a $O/k1bench.txt
K1BENCH_R=192 timeout 300 python tools/k1bench.py fin48 2100 18250 2>&1 | grep k1bench | tee -a $O/k1bench.txt
