#!/bin/bash
# round 6, GPU pass f: final validation (fanout sum, fused cross-attention keys on top of pass r):
# whole GPU suite, smoke, bench lines of every config (LoRA r = 64 / 8 replayed, BART replayed at the full batch, emulated ranks)
O=gpurun_out/r6f; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --graph off --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_eager.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
timeout 600 python bench.py --model lora --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>&1
timeout 600 python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_video.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6f/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "peak GB", j.get("peak_memory_GB"), "frac", j["roofline"]["frac"], "op_us", j["roofline"].get("op_avg_us"),
                  {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k5_fwd", "k5_bwd", "k4_fwd", "k4_wgrad", "k2_fwd", "k2_bwd", "k3_fwd", "k3_bwd") if n in k})
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1200:])
P
