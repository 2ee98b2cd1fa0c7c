#!/bin/bash
O=gpurun_out/r4as; mkdir -p $O
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for f in 1 2; do for d in 0 1 2 3; do
  echo "# VLPET_ATTN_BWD2=$f VLPET_DBG=$d" >> $O/attnbench_ab.txt
  VLPET_DBG=$d VLPET_ATTN_BWD2=$f timeout 300 python tools/attnbench.py 2>&1 | grep -E "vqa|caption" | cut -c1-75 >> $O/attnbench_ab.txt
done; done
cat $O/attnbench_ab.txt
