#!/bin/bash
# round 2, pass bm: rocprofv3 kernel statistics of the T5 (configs[2]) and LoRA (configs[3]) benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bm; mkdir -p $O
for m in t5 lora; do
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$m -o $m -- python $GRAFT_REPO_ROOT/bench.py --model $m --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof_$m.log 2>&1 )
f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${m}_kernel_stats.csv
rm -rf $O/prof_$m
done
