#!/bin/bash
# round 2, pass af: fused q|k|v projection + strided attention entry points: parity, bench A/B, GEMM table for the new shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2af; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_attention.py tests/test_host_golden.py tests/test_gpu_modules.py tests/test_gpu_dp.py -m gpu -q -x > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sub.log
tail -4 $O/pytest_sub.log | cut -c1-250
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_fused_untuned.json.log 2>$O/bench_a.err
VLPET_NO_FUSED_QKV=1 timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_separate.json.log 2>$O/bench_b.err
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=8
export PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20
timeout 1500 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_bart.log 2>&1
cp gpurun_out/tunableop_gfx950_new.csv $O/tunableop_gfx950.csv; cp gpurun_out/tunableop_gfx950_new.csv vl-pet_amd/tuning/tunableop_gfx950.csv
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_fused_tuned.json.log 2>$O/bench_c.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2af/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j.get("backbone_gemm_table"))
    except Exception as e: print(f, "ERR", e)
PY
wc -l $O/tunableop_gfx950.csv
