#!/bin/bash
mkdir -p gpurun_out/r4an
O=gpurun_out/r4an
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_stamps.so K1BENCH_FWD_ONLY=1
VLPET_FWD2P=2 K1BENCH_R=96 timeout 120 python tools/k1bench.py stamps 28000 2>&1 | grep -E "f2 stamps" | tail -12 > $O/passA_r96_28000.txt
VLPET_FWD2P=2 K1BENCH_R=192 timeout 120 python tools/k1bench.py stamps 18250 2>&1 | grep -E "f2 stamps" | tail -12 > $O/passA_r192_18250.txt
VLPET_FWD2P=3 K1BENCH_R=192 timeout 120 python tools/k1bench.py stamps 18250 2>&1 | grep -E "f2 stamps" | tail -8 > $O/passB_r192_18250.txt
python bench.py --emulate-ranks 8 --no-cpu-baseline --steps 8 --warmup 2 > $O/bench_r8.json.log 2> $O/err.txt
