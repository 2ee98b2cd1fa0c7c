#!/bin/bash
# round 6, GPU pass i: after the K4 workspace fix (one area per device, never reallocated): LoRA / video / emulated-rank bench lines, the lowrank and graph suites
O=gpurun_out/r6i; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1200 python -m pytest tests/test_gpu_lowrank.py tests/test_gpu_graph.py tests/test_gpu_k4.py tests/test_gpu_dp.py -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest.txt
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bart_2ranks_gloo_1gpu.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6i/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "peak GB", j.get("peak_memory_GB"), "frac", j["roofline"]["frac"], "n_gpus", j["n_gpus"],
                  {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k5_fwd", "k5_bwd", "k4_fwd", "k2_bwd", "k3_fwd", "k3_bwd") if n in k})
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
