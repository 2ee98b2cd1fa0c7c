#!/bin/bash
# round 6, GPU pass e: whole GPU suite + smoke + bench lines (bart default, t5) after the pruning / fallbacks / reduce-scatter policy
O=gpurun_out/r6e; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6e/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], "median ms", j["step_ms_median"], "steady", j["steady_state"]["value"], "op_us", j["roofline"].get("op_avg_us"), "frac", j["roofline"]["frac"],
                  {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k2_bwd", "k5_fwd", "k5_bwd", "k4_fwd") if n in k})
    if not ok: print(f, "NO JSON", open(f).read()[-800:])
P
