#!/bin/bash
O=gpurun_out/r4bh; mkdir -p $O
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for cap in 128 256 384 512 768; do
  VLPET_DBG=$cap timeout 200 python tools/k5abi.py 1240 2100 3500 5880 10000 28000 2>&1 | grep k5abi | sed "s/^/cap=$cap /" >> $O/k5cap.txt
done
cat $O/k5cap.txt
