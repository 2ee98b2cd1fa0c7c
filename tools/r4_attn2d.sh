#!/bin/bash
O=$PWD/gpurun_out/${OUT:-r4aw}; mkdir -p $O
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for cfg in ${CFGS:-0,0 1,0 2,0 1,2 2,2}; do
  a=${cfg%,*}; d=${cfg#*,}
  VLPET_ATTN_BWD2=$a VLPET_DBG=$d timeout 120 python tools/attnbwd_bench.py "bwd2=$a dbg=$d" 2>&1 | grep attnbwd >> $O/attnbwd.txt
done
cat $O/attnbwd.txt
