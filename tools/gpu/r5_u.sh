#!/bin/bash
# round 5, GPU pass u: timing ablations of pet_cols6y.hip (results wrong on purpose): what a step waits for
O=gpurun_out/r5u; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for v in "" _c6y_abl1 _c6y_abl2 _c6y_abl4; do
  echo "== lib$v" | tee -a $O/k1bench.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so K1BENCH_R=192 python tools/k1bench.py "c6y$v" 18250 28000 2>&1 | grep -v amdgpu.ids | sed 's/| previous split.*| default://' | tee -a $O/k1bench.txt
done
done
