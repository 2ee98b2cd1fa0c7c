#!/bin/bash
# round 2, pass an: f4 timings (per launch, vs the library composition, same box) + rocprofv3 kernel stats of the same command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2an; mkdir -p $O
timeout 600 python tools/f4bench.py 18700 28000 2>&1 | grep -v amdgpu.ids | tee $O/f4bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o f4 -- python $GRAFT_REPO_ROOT/tools/f4bench.py 18700 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats*" | head; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -c1-200 | tee $O/f4_kernel_stats_head.csv
find $O/prof -name "*.db" -delete; find $O/prof -name "*trace.csv" -size +8M -delete
