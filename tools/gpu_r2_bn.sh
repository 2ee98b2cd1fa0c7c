#!/bin/bash
# round 2, pass bn: T5 host with the RMS norm returning the activation dtype (the encoder ran in fp32 before): T5 tests, T5 bench
# + rocprofv3 kernel statistics
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bn; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -k "t5 or T5 or wide or host_golden or k4" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest_t5.txt
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/t.err; tail -3 $O/t.err
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_t5 -o t5 -- python $GRAFT_REPO_ROOT/bench.py --model t5 --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof_t5.log 2>&1 )
f=$(find $O/prof_t5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/t5_kernel_stats.csv
rm -rf $O/prof_t5
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r2bn/bench_t5.json.log").read().strip().splitlines()[-1]); r=j["roofline"]
print(j["value"], j["ms_per_step"], r)
for k,v in j["kernels"].items(): print("  ",k,v["launches"],v["avg_us"],v.get("hbm_frac"))
PY
