#!/bin/bash
mkdir -p gpurun_out/r4ah
O=gpurun_out/r4ah
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_tail.py tests/test_gpu_k4.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2; do
python bench.py --emulate-ranks 8 --no-cpu-baseline --kernel-table off > $O/bart_r8_plain_$rep.json.log 2>> $O/err.txt
python bench.py --emulate-ranks 8 --no-cpu-baseline --kernel-table off --side-finalize > $O/bart_r8_sidefin_$rep.json.log 2>> $O/err.txt
python bench.py --no-cpu-baseline --kernel-table off > $O/bart_plain_$rep.json.log 2>> $O/err.txt
python bench.py --no-cpu-baseline --kernel-table off --side-finalize > $O/bart_sidefin_$rep.json.log 2>> $O/err.txt
done
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline --kernel-table off > $O/t5_r8_plain.json.log 2>> $O/err.txt
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline --kernel-table off --side-finalize > $O/t5_r8_sidefin.json.log 2>> $O/err.txt
tail -3 $O/err.txt
