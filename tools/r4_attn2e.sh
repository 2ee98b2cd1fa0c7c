#!/bin/bash
O=$PWD/gpurun_out/r4bc; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -3
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for cfg in 0,0; do
  IFS=, read d occ <<< "$cfg"
  VLPET_DBG=$d VLPET_ATTN_OCC=$occ timeout 120 python tools/attnbwd_bench.py "dbg=$d occ=$occ" 2>&1 | grep attnbwd >> $O/attnbwd.txt
done
cat $O/attnbwd.txt
