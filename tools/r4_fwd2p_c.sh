#!/bin/bash
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
export HIP_FORCE_DEV_KERNARG=1 K1BENCH_FWD_ONLY=1 VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
echo "== parity of the variants" | tee $O/log.txt
for v in "1 0" "0 1"; do set -- $v
VLPET_FWD2P_A=$1 VLPET_FWD2P_B=$2 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/log.txt
done
for R in 96 192; do
for v in "2 0 0" "2 1 0" "3 0 0" "3 0 1" "1 0 0" "1 1 1" "0 0 0"; do set -- $v
  VLPET_FWD2P=$1 VLPET_FWD2P_A=$2 VLPET_FWD2P_B=$3 K1BENCH_R=$R timeout 300 python tools/k1bench.py "mode$1-A$2-B$3" 2100 3500 15272 18250 28000 33200 46648 2>&1 | grep k1bench | tee -a $O/k1fwd_r$R.txt
done
done
