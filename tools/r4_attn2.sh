#!/bin/bash
# persistent attention backward: parity (product + diagnosis library) and the same-box A/B against the one-pair kernel
O=gpurun_out/r4ao; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py -x -q -m gpu > $O/pytest_attn.txt 2>&1
tail -5 $O/pytest_attn.txt
VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -m gpu -k "persistent or six_wave" > $O/pytest_attn_dbg.txt 2>&1
tail -3 $O/pytest_attn_dbg.txt
for f in 0 1; do
  echo "# VLPET_ATTN_BWD2=$f" >> $O/attnbench_ab.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_ATTN_BWD2=$f timeout 300 python tools/attnbench.py 2>&1 | grep -E "B=" >> $O/attnbench_ab.txt
done
cat $O/attnbench_ab.txt
