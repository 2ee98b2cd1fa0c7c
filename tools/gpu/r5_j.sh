#!/bin/bash
# round 5, GPU pass j: timing ablations of pass 2 of the K1 backward (results wrong on purpose): what would pipelining the up-side products /
# the down-side finish / the elementwise block buy at most?
O=gpurun_out/r5j; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for lib in head abl1 abl2 abl3 abl4 abl7; do
  if [ $lib = head ]; then unset VLPET_LIB; else export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_$lib.so; fi
  python tools/k1bench.py $lib 28000 2>&1 | grep k1bench | sed 's/previous split.*default/default/'
  K1BENCH_COLD=1 python tools/k1bench.py $lib 28000 2>&1 | grep k1bench | sed 's/previous split.*default/default/'
done; done 2>&1 | tee $O/k1bench_abl.txt
