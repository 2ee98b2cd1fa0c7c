#!/bin/bash
# round 2, pass r: attention kernels restructured (forward: a wave per (batch, head); backward: (tile, d half) units)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
O=gpurun_out/r2s
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q > $O/pytest_attn.log 2>&1; echo "rc=$?" >> $O/pytest_attn.log; tail -5 $O/pytest_attn.log | cut -c1-250
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_EAGER_ATTENTION=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_sdpa.json.log 2>$O/bench_bart_sdpa.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2s/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"]); k=j.get("kernels",{}); print({n:(v["avg_us"],v.get("hbm_frac")) for n,v in k.items() if ("attn" in n)})
    except Exception as e: print(f, "ERR", e)
PY
