#!/bin/bash
O=gpurun_out/r4ar; mkdir -p $O
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | tail -2
for P in 0.1; do for f in 0 1 2; do
  echo "# VLPET_ATTN_BWD2=$f p=$P" >> $O/attnbench_ab.txt
  ATTNBENCH_P=$P VLPET_ATTN_BWD2=$f timeout 300 python tools/attnbench.py 2>&1 | grep -E "B=" | cut -c1-75 >> $O/attnbench_ab.txt
done; done
cat $O/attnbench_ab.txt
