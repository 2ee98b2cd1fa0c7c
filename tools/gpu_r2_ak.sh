#!/bin/bash
# round 2, pass ak: validation of the second session's state: full GPU suite, smoke, every bench line, dp2 (gloo, strong), default bench with the cpu baseline (timed)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ak; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log | cut -c1-300
t0=$(date +%s)
timeout 900 python bench.py > $O/bench_bart_default.json.log 2>$O/bench_bart.err
echo "default bench (with cpu baseline): $(( $(date +%s) - t0 )) s"; grep "cpu_baseline\]" $O/bench_bart.err | cut -c1-200
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>$O/bench_lora8.err
timeout 400 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_video.json.log 2>$O/bench_video.err
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/bench_t5.err
timeout 400 python bench.py --gpus 2 --backend gloo --scaling strong --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_strong.json.log 2>$O/bench_dp2.err
timeout 400 python bench.py --gpus 2 --backend gloo --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_weak.json.log 2>$O/bench_dp2w.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ak/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["n_gpus"], j["roofline"]["frac"], j["roofline"].get("op_frac")); 
        if "cpu_baseline" in j: print("   cpu:", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["cpu_baseline"]["thread_sweep"])
    except Exception as e: print(f, "ERR", e)
PY
