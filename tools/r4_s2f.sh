#!/bin/bash
mkdir -p gpurun_out/r4r
O=gpurun_out/r4r
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r8 -o kt -- python bench.py --emulate-ranks 8 --no-cpu-baseline --kernel-table off --steps 40 > $O/bench_r8.json.log 2> $O/bench_r8.err
find $O/prof_r8 -name "kt_kernel_stats.csv" -exec cp {} $O/kernel_stats_bart_rank1of8_graph.csv \;
rm -rf $O/prof_r8
