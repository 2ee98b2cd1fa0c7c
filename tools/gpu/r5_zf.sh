#!/bin/bash
# round 5, GPU pass zf: last validation at HEAD -- whole GPU suite, smoke, default bench line, T5 line
O=gpurun_out/r5zf; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_default.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
python - <<'P'
import json
for f in ("bench_default", "bench_t5"):
    for l in open(f"gpurun_out/r5zf/{f}.json.log"):
        if l.startswith("{"):
            j = json.loads(l); print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("op_avg_us"))
P
