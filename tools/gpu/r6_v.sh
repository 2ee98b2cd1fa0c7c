#!/bin/bash
# round 6, GPU pass v: rocprofv3 kernel trace of the T5 and LoRA bench commands (replayed), for a per-step breakdown
O=gpurun_out/r6v; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for m in t5 lora; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o kt -- python bench.py --model $m --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_${m}_under_rocprofv3.json.log 2>&1
find $O/prof_$m -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$m.csv \;
find $O/prof_$m -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_$m.csv \;
rm -rf $O/prof_$m
done
ls -la $O
