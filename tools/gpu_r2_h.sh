#!/bin/bash
# round 2, pass h: full suite + the four bench lines with the residual link and the batched re-pack
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
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
for f in sorted(glob.glob("gpurun_out/r2h/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["n_gpus"])
    except Exception as e: print(f, "ERR", e)
PY
