#!/bin/bash
# round 2, pass g: full GPU suite (residual link, dx1 accumulation), bench A/B of the link, SDPA backend experiment
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 300 python tools/kbench.py 28000 bf16 > $O/kbench_28000.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_NO_LINK=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_nolink.json.log 2>$O/bench_bart_nolink.err
for b in flash efficient math; do VLPET_SDPA=$b timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_bart_sdpa_$b.json.log 2>$O/bench_bart_sdpa_$b.err; done
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/bench_t5.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2g/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
