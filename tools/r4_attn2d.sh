#!/bin/bash
# attention kernels through the C ABI alone (tools/attnbwd_bench.py), diagnosis build: the shipped kernels and the backward's ablations
# (VLPET_DBG: 1 = loads only, 4 = loads + compute, no stores, 8 = loads + stores of zeros, no compute) -> profiles/r04_attnbwd_ablation.txt
O=$PWD/gpurun_out/${OUT:-r4attn}; mkdir -p $O
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for d in 0 1 4 8; do
  VLPET_DBG=$d timeout 120 python tools/attnbwd_bench.py "dbg=$d" 2>&1 | grep attnbwd >> $O/attnbwd.txt
done
cat $O/attnbwd.txt
