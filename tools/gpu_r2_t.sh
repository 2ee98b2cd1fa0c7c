#!/bin/bash
# round 2, pass t: full suite + kernel profile with the attention kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
O=gpurun_out/r2t
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r2t/prof_bart/bart_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "attn_" in n:
        agg[n[:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v.sort(); print(k, len(v), "min %.1f med %.1f p90 %.1f max %.1f sum %.1f ms" % (v[0], v[len(v)//2], v[int(len(v)*0.9)], v[-1], sum(v)/1e3))
    print("   ", [round(x) for x in v[::max(1,len(v)//24)]])
PY
find $O -name "*_kernel_trace.csv" -delete
