#!/bin/bash
# round 5, GPU pass zb: feature blocks of pass 1 at the per-rank sizes (debug build, VLPET_DZ2_FSPLIT = 4 (default) / 6 / 12)
O=gpurun_out/r5zb; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for f in 4 6 12 2; do
  echo "== VLPET_DZ2_FSPLIT=$f" | tee -a $O/k1bench.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_DZ2_FSPLIT=$f K1BENCH_R=192 python tools/k1bench.py "r192 f$f" 1100 2128 3500 6000 2>&1 | grep -v amdgpu.ids | sed 's/| previous split.*| default://' | tee -a $O/k1bench.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_DZ2_FSPLIT=$f python tools/k1bench.py "r96 f$f" 1932 3500 5880 8000 2>&1 | grep -v amdgpu.ids | sed 's/| previous split.*| default://' | tee -a $O/k1bench.txt
done
