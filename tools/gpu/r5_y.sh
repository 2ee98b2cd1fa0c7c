#!/bin/bash
# round 5, GPU pass y: whole GPU suite + smoke + the four bench configs + emulated ranks at HEAD (second session)
O=gpurun_out/r5y; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_video.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
python - <<'P' | tee gpurun_out/r5y/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5y/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "frac", j["roofline"]["frac"], "op_us", j["roofline"].get("op_avg_us"),
                  {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k5_fwd", "k5_bwd", "k4_fwd", "k2_bwd", "k3_fwd", "k3_bwd") if n in k})
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
