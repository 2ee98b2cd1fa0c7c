#!/bin/bash
# round 2, pass aa: bench with the live brackets restricted to the roofline op vs every launch; full GPU suite; dp2 gloo
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aa; mkdir -p $O
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_after.json.log 2>$O/bench_after.err
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table inline > $O/bench_inline.json.log 2>$O/bench_inline.err
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_off.json.log 2>$O/bench_off.err
timeout 400 python bench.py --gpus 2 --backend gloo --scaling strong --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_strong.json.log 2>$O/bench_dp2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2aa/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["n_gpus"], j["roofline"]["frac"], j["roofline"].get("op_frac"), j["roofline"].get("op_avg_us")); print({n:v["avg_us"] for n,v in j.get("kernels",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
