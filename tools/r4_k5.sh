#!/bin/bash
mkdir -p gpurun_out/r4h
O=gpurun_out/r4h
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_optim.py tests/test_gpu_k4.py tests/test_gpu_lowrank.py -m gpu -q 2>&1 | tail -5 | tee $O/log.txt
for rep in 1 2; do
for pre in 1 0; do
for M in 15272 28000 30384 46648; do
K5BENCH_PRENORM=$pre python tools/k5bench.py $M 2>&1 | grep -v "^SAVE" | sed "s/^/prenorm=$pre /" | tee -a $O/k5bench.txt
done; done; done
timeout 900 python -m pytest tests/test_gpu_cols.py -m gpu -q -s 2>&1 | grep -E "element-wise|passed|failed" | tail -40 | tee -a $O/elwise.txt
