#!/bin/bash
mkdir -p gpurun_out/r4x
O=gpurun_out/r4x
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for occ in 2 3; do VLPET_ATTN_OCC=$occ timeout 300 python tools/attnbench_t5.py occ=$occ 2>&1 | grep attnbench_t5 | tee -a $O/attnbench_t5.txt; done
unset VLPET_LIB
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
python bench.py --model t5 --no-cpu-baseline > $O/bench_t5.json.log 2> $O/bench_t5.err
