#!/bin/bash
mkdir -p gpurun_out/r4al
O=gpurun_out/r4al
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_host_golden.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2; do
VLPET_AB=1 VLPET_NO_FUSED_QKV=1 python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline --kernel-table off > $O/t5_r8_sepqkv_$rep.json.log 2>> $O/err.txt
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline --kernel-table off > $O/t5_r8_fusedqkv_$rep.json.log 2>> $O/err.txt
VLPET_AB=1 VLPET_NO_FUSED_QKV=1 python bench.py --model t5 --no-cpu-baseline --kernel-table off > $O/t5_sepqkv_$rep.json.log 2>> $O/err.txt
python bench.py --model t5 --no-cpu-baseline --kernel-table off > $O/t5_fusedqkv_$rep.json.log 2>> $O/err.txt
done
tail -2 $O/err.txt
