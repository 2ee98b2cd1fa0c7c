#!/bin/bash
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
rm -f $O/abl_*.txt
export HIP_FORCE_DEV_KERNARG=1 K1BENCH_FWD_ONLY=1
for R in 96 192; do
for mode in 2 3; do
for v in dbg abl1 abl2 abl4 abl7; do
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_$v.so VLPET_FWD2P=$mode K1BENCH_R=$R timeout 300 python tools/k1bench.py "mode$mode-$v" 3500 18250 28000 2>&1 | grep k1bench | tee -a $O/abl_r$R.txt
done; done; done
