#!/bin/bash
# round 2, pass q: short-sequence attention in the BART host: full suite + bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
O=gpurun_out/r2q
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log | cut -c1-200
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_EAGER_ATTENTION=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_sdpa.json.log 2>$O/bench_bart_sdpa.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2q/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"]); k=j.get("kernels",{}); print({n:(v["avg_us"],v.get("hbm_frac")) for n,v in k.items() if ("attn" in n)})
    except Exception as e: print(f, "ERR", e)
PY
