#!/bin/bash
mkdir -p gpurun_out/r4ak
O=gpurun_out/r4ak
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_stamps.so K1BENCH_FWD_ONLY=1
for M in 15272 28000; do
for mode in 2 3; do
VLPET_FWD2P=$mode K1BENCH_R=96 timeout 120 python tools/k1bench.py stamps-mode$mode $M 2>&1 | grep -E "f2 stamps|k1bench" | sort | uniq -c | sort -rn | head -12 | tee -a $O/stamps_r96.txt
done; done
VLPET_FWD2P=2 K1BENCH_R=192 timeout 120 python tools/k1bench.py stamps-mode2 18250 2>&1 | grep -E "f2 stamps|k1bench" | sort | uniq -c | sort -rn | head -12 | tee -a $O/stamps_r192.txt
