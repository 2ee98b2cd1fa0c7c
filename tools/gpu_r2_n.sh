#!/bin/bash
# round 2, pass n: fused LM-head cross entropy: tests, full suite, bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
O=gpurun_out/r2n
timeout 600 python -m pytest tests/test_gpu_act.py tests/test_gpu_loss.py -m gpu -q -x > $O/pytest_act.log 2>&1; echo "rc=$?" >> $O/pytest_act.log; tail -15 $O/pytest_act.log
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_EAGER_LM_LOSS=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_eager_loss.json.log 2>$O/bench_bart_eager.err
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/bench_t5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2n/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"]); k=j.get("kernels",{}); print({n:(v["avg_us"],v.get("hbm_frac")) for n,v in k.items() if ("ffn" in n or "ce_" in n)})
    except Exception as e: print(f, "ERR", e)
PY
