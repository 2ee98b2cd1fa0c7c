#!/bin/bash
# round 5, GPU pass za: the N > 1 launcher path of bench.py on one GPU (two gloo ranks sharing it; weak and strong scaling), sanity after the session's changes
O=gpurun_out/r5za; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_weak.json.log 2>&1; tail -c 1500 $O/bench_dp2_gloo_weak.json.log
timeout 900 python bench.py --gpus 2 --backend gloo --scaling strong --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_dp2_gloo_strong.json.log 2>&1; tail -c 1200 $O/bench_dp2_gloo_strong.json.log
