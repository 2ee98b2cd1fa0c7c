#!/bin/bash
# round 2, pass bb: rocprofv3 kernel statistics of the default bench with the GEMM gradient hand-over off / on (same box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bb; mkdir -p $O
for mode in nolink link; do
  if [ $mode = nolink ]; then export VLPET_NO_GEMM_LINK=1; else unset VLPET_NO_GEMM_LINK; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$mode -o bart -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof_$mode.log 2>&1 )
  f=$(find $O/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${mode}_kernel_stats.csv
  find $O/prof_$mode -name "*trace.csv" -delete
done
rm -rf $O/prof_nolink $O/prof_link
tail -2 $O/prof_link.log
