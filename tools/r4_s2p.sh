#!/bin/bash
mkdir -p gpurun_out/r4ab
O=gpurun_out/r4ab
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in nodefer defer; do
  if [ $v = nodefer ]; then export VLPET_NO_DEFER_REDUCES=1; else unset VLPET_NO_DEFER_REDUCES; fi
  VLPET_AB=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o kt -- python bench.py --model lora --no-cpu-baseline --kernel-table off --steps 12 > $O/bench_$v.json.log 2> $O/bench_$v.err
  find $O/prof_$v -name "kt_kernel_stats.csv" -exec cp {} $O/kernel_stats_lora_$v.csv \;
  rm -rf $O/prof_$v
done
