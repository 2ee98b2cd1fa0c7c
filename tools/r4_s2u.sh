#!/bin/bash
mkdir -p gpurun_out/r4ag
O=gpurun_out/r4ag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o kt -- python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline --kernel-table off --steps 40 > $O/bench.json.log 2> $O/bench.err
find $O/prof -name "kt_kernel_stats.csv" -exec cp {} $O/kernel_stats_t5_rank1of8_graph.csv \;
rm -rf $O/prof
