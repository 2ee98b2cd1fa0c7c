#!/bin/bash
# round 6, GPU pass q: rocprofv3 kernel statistics of the default bench command (replayed graphs + the eager region), csv
O=gpurun_out/r6q2; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o kt -- python bench.py --no-cpu-baseline > $O/bench_bart_under_rocprofv3.json.log 2>&1
find $O/prof_bart -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bart.csv \;
find $O/prof_bart -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_bart.csv \;
python - <<'P'
# per-step view from the trace: keep the kernel launches of the LAST replayed vqa step only would need markers; instead write the trace's
# totals per kernel name restricted to the timed region's duration is not possible without markers -> keep the stats csv, drop the trace if large
import os
p = "gpurun_out/r6q2/kernel_trace_bart.csv"
if os.path.exists(p) and os.path.getsize(p) > 40e6:
    os.remove(p)
P
rm -rf $O/prof_bart
ls -la $O
