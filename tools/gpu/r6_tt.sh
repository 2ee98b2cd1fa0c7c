#!/bin/bash
# round 6, GPU pass tt: bench lines with the extended TunableOp table (the fused key projection's shapes and T5's 2- / 4-rank shapes measured)
O=gpurun_out/r6tt; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --gemm-table off --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_bart_notable.json.log 2>&1
timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_bart_b.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 40 --warmup 6 --no-cpu-baseline > $O/bench_r8.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 2 --steps 40 --warmup 6 --no-cpu-baseline > $O/bench_r2.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 4 --steps 40 --warmup 6 --no-cpu-baseline > $O/bench_r4.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6tt/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l)
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], j["backbone_gemm_table"])
P
