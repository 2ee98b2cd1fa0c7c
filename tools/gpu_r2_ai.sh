#!/bin/bash
# round 2, pass ai: bench + kernel profile of the current state
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ai; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
python - <<'PY'
import json,csv,glob
j=json.loads(open("gpurun_out/r2ai/bench_bart.json.log").read().strip().splitlines()[-1]); print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("op_frac"), j["roofline"].get("op_avg_us")); print({n:v["avg_us"] for n,v in j.get("kernels",{}).items()})
f=glob.glob("gpurun_out/r2ai/prof_bart/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step (12 steps):", tot/1e6/12)
for r in rows[:40]:
    print(f"{float(r['Percentage']):5.2f}% {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f}us  {r['Name'][:100]}")
PY
