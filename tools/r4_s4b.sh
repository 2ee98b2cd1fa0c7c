#!/bin/bash
# rocprofv3 kernel statistics of the bench lines at HEAD (same command as the bench line, --kernel-table off, 12 steps)
O=$PWD/gpurun_out/${OUT:-r4prof}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
R=/root/repo
for m in ${MODELS:-bart}; do
  ( cd $R; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o kt -- python bench.py --model $m --no-cpu-baseline --kernel-table off --steps 12 > $O/bench_$m.json.log 2> $O/bench_$m.err )
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_$m.csv
  rm -rf $O/prof_$m
done
ls -la $O
