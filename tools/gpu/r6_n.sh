#!/bin/bash
# round 6, GPU pass n: bench.py default switched to replayed graphs for the headline workload (eager region after the timed one
# carries the roofline brackets): stability (three default runs + a 100-step run), the eager line beside it, rocprofv3 stats of the default command
O=gpurun_out/r6n; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for i in 1 2 3; do timeout 600 python bench.py $( [ $i != 1 ] && echo --no-cpu-baseline ) > $O/bench_bart_$i.json.log 2>&1; done
timeout 600 python bench.py --graph off --no-cpu-baseline > $O/bench_bart_eager.json.log 2>&1
timeout 600 python bench.py --steps 100 --warmup 4 --no-cpu-baseline > $O/bench_bart_100.json.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $O/bench_bart_4.json.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bart -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_bart_under_rocprofv3.json.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_bart.csv \;
find $O/prof -type f ! -name "*kernel_stats.csv" -delete
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6n/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "eager_region", j.get("eager_region", {}).get("ms_per_step"), "frac", j["roofline"]["frac"], "op_us", j["roofline"].get("op_avg_us"))
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1200:])
P
