#!/bin/bash
# round 2, pass ar: cold-cache microbenchmarks of K1 (the in-step condition), and the two-pass backward (7 units of traffic
# instead of 11; rejected on warm microbenchmarks) A/B-ed IN THE STEP: bench default vs VLPET_BWD3=1, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ar; mkdir -p $O
{
for M in 15272 28000 31616 46648; do
  K1BENCH_COLD=1 timeout 200 python tools/k1bench.py split $M
  K1BENCH_COLD=1 VLPET_BWD3=1 timeout 200 python tools/k1bench.py twopass $M
done
timeout 120 python tools/k1bench.py warm 28000
} 2>&1 | grep -v amdgpu.ids | tee $O/k1bench_cold.txt
for i in 1 2; do
  timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --kernel-table off > $O/bench_split_$i.json.log 2>$O/a$i.err
  VLPET_BWD3=1 timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --kernel-table off > $O/bench_twopass_$i.json.log 2>$O/b$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ar/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]; print(f.split('/')[-1], j["value"], j["ms_per_step"], "op_avg_us", r.get("op_avg_us"), "kernel", r.get("kernel"))
    except Exception as e: print(f, "ERR", e)
PY
