#!/bin/bash
# round 2, pass bi: validation after the activation-pass change: whole GPU suite, smoke, default bench + rocprofv3 statistics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bi; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 > $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench.err
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bart -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bart_kernel_stats.csv
rm -rf $O/prof
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2bi/bench_default.json.log").read().strip().splitlines()[-1]); r=j["roofline"]
print(j["value"], j["ms_per_step"], r["frac"], r["op_frac"], j["kernels"].get("ffn_act_fwd"), j["kernels"].get("ffn_act_bwd"))
PY
