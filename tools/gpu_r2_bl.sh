#!/bin/bash
# round 2, pass bl: in-step effect of the attention-backward occupancy change: rocprofv3 kernel statistics + bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bl; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bart -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bart_kernel_stats.csv
rm -rf $O/prof
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench.json.log 2> $O/b.err
VLPET_ATTN_OCC=2 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_occ2.json.log 2> $O/b2.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_b.json.log 2> $O/b.err
VLPET_ATTN_OCC=2 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_occ2_b.json.log 2> $O/b2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bl/bench*.json.log")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
PY
