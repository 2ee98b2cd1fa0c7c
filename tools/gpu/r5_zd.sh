#!/bin/bash
# round 5, GPU pass zd: the LDS-DMA + barrier floor of pet_cols.hip (r = 96 pass 2; -DVLPET_COLS_ABL=8: no role work between the barriers)
O=gpurun_out/r5zd; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for v in "" _cols_abl8; do
  echo "== lib$v" | tee -a $O/k1bench.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so python tools/k1bench.py "cols$v" 15272 28000 31616 2>&1 | grep -v amdgpu.ids | sed 's/| previous split.*| default://' | tee -a $O/k1bench.txt
done
done
