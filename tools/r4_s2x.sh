#!/bin/bash
mkdir -p gpurun_out/r4aj
O=gpurun_out/r4aj
for m in t5 lora video; do
python bench.py --model $m --no-cpu-baseline > $O/bench_${m}_graph.json.log 2>> $O/err.txt
python bench.py --model $m --no-cpu-baseline --graph off > $O/bench_${m}_eager.json.log 2>> $O/err.txt
done
tail -3 $O/err.txt
