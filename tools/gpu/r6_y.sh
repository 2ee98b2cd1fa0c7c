#!/bin/bash
# round 6, GPU pass y: kernel trace of the emulated rank-1-of-8 step (BART), per-step breakdown
O=gpurun_out/r6y; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o kt -- python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_rank1of8_under_rocprofv3.json.log 2>&1
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_rank1of8.csv \;
rm -rf $O/prof
python tools/step_breakdown.py $O/kernel_trace_rank1of8.csv $O/bench_rank1of8_under_rocprofv3.json.log 60 | tee $O/breakdown.txt
