#!/bin/bash
# round 5, GPU pass m: per-kernel TIMELINE (not only statistics) of the emulated rank-1-of-8 step in graph mode -- where the 5.6 ms go
O=gpurun_out/r5m; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in bart t5; do
  extra=""; [ $m = t5 ] && extra="--model t5"
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof_$m -o kt -- python bench.py $extra --emulate-ranks 8 --steps 6 --warmup 3 --kernel-table off --no-cpu-baseline > $O/bench_${m}_trace.log 2>&1
  f=$(find $O/prof_$m -name "kt_kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" $O/trace_$m.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-12000:]                      # the last steps only
with open(sys.argv[2], "w") as f:
    f.write("start_ns,dur_ns,grid,wg,name\n")
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        f.write(f'{int(r["Start_Timestamp"]) - t0},{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])},{r.get("Grid_Size_X", r.get("Grid_Size", ""))},{r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))},"{r["Kernel_Name"][:120]}"\n')
P
  rm -rf $O/prof_$m
done
ls -la $O
