#!/bin/bash
# round 6, GPU pass a: K4 statistics exchange under held CUs (repair path), whole K4 suite, default bench line with step_ms
O=gpurun_out/r6a; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py -x -q 2>&1 | tail -15 | tee $O/pytest_k4.txt
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json
for l in open("gpurun_out/r6a/bench_bart.json.log"):
    if l.startswith("{"):
        j = json.loads(l); k = j["kernels"]
        print(j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "slow", j["slow_steps"])
        print("step_ms", j["step_ms"]); print("settling", j["settling_rounds_ms"]); print("by task", j["step_ms_by_task"])
        print("frac", j["roofline"]["frac"], {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k1_bwd_op", "k5_fwd", "k5_bwd", "k4_fwd", "k4_wgrad", "k2_fwd", "k2_bwd") if n in k})
P
tail -5 $O/bench_bart.json.log | cut -c1-600
