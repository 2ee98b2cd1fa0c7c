#!/bin/bash
# round 6, GPU pass p: the position-branch kernel after the multi-row trips + the parallel finalize: tests, micro-benchmark, bench line
O=gpurun_out/r6p; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py -q -x -k "position or frozen_token" 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/pytest_pos.txt
timeout 300 python tools/vispos_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/vispos_bench.txt
timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6p/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"],
                  {n: k[n]["avg_us"] for n in ("k4_fwd", "k4_ln_bwd", "k4_wgrad", "k4_pos_fwd", "k4_pos_bwd") if n in k})
P
