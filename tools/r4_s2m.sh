#!/bin/bash
mkdir -p gpurun_out/r4y
O=gpurun_out/r4y
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in video; do
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o kt -- python bench.py --model $m --no-cpu-baseline --kernel-table off --steps 12 > $O/bench_$m.json.log 2> $O/bench_$m.err
find $O/prof_$m -name "kt_kernel_stats.csv" -exec cp {} $O/kernel_stats_bench_$m.csv \;
rm -rf $O/prof_$m
done
