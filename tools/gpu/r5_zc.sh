#!/bin/bash
# round 5, GPU pass zc: final validation at HEAD -- whole GPU suite, smoke, default bench line, T5 emulated rank
O=gpurun_out/r5zc; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json.log 2>&1; tail -c 600 $O/bench_default.json.log
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
python - <<'P'
import json
for f in ("bench_default", "bench_t5_rank1of8"):
    for l in open(f"gpurun_out/r5zc/{f}.json.log"):
        if l.startswith("{"):
            j = json.loads(l); print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline"))
P
