#!/bin/bash
mkdir -p gpurun_out/r4aa
O=gpurun_out/r4aa
for rep in 1 2; do
VLPET_AB=1 VLPET_NO_DEFER_REDUCES=1 python bench.py --model lora --no-cpu-baseline --kernel-table off > $O/bench_lora_nodefer_$rep.json.log 2> $O/err.txt
VLPET_AB=1 python bench.py --model lora --no-cpu-baseline --kernel-table off > $O/bench_lora_defer_$rep.json.log 2>> $O/err.txt
done
