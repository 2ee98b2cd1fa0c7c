#!/bin/bash
mkdir -p gpurun_out/r4ae
O=gpurun_out/r4ae
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_cols.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so K1BENCH_R=192
for v in 1 2 4; do
VLPET_DZ2_FSPLIT=$v VLPET_DZ6=2 timeout 300 python tools/k1bench.py dz6-fsplit=$v 1200 2128 3528 5000 8192 2>&1 | grep k1bench | tee -a $O/k1bench_r192_small.txt
done
VLPET_DZ6=0 timeout 300 python tools/k1bench.py chain-split 1200 2128 3528 5000 8192 2>&1 | grep k1bench | tee -a $O/k1bench_r192_small.txt
unset VLPET_LIB
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline > $O/bench_t5_rank1of8_graph.json.log 2> $O/bench_t5_rank1of8.err
