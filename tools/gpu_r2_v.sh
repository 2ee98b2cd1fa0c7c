#!/bin/bash
# round 2, pass v: full suite, smoke, the four bench lines, dp2 gloo, kernel profile (after the backbone passes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
O=gpurun_out/r2v
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>$O/bench_bart.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_video.json.log 2>$O/bench_video.err
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/bench_t5.err
timeout 400 python bench.py --gpus 2 --backend gloo --scaling strong --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_strong.json.log 2>$O/bench_dp2.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2v/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["n_gpus"]); k=j.get("kernels",{}); print({n:(v["avg_us"],v.get("hbm_frac")) for n,v in k.items() if ("attn" in n or "ce_" in n or "ffn" in n)})
    except Exception as e: print(f, "ERR", e)
PY
