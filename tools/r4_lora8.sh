#!/bin/bash
O=gpurun_out/r4bl; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ng.py -x -q -m gpu -k "k3" 2>&1 | tail -5
for M in 2500 28000; do timeout 300 python tools/k3bench.py $M 2>&1 | grep "r=   8" | sed "s/^/M=$M /" >> $O/k3bench_r8.txt; done
cat $O/k3bench_r8.txt
