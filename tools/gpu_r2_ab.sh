#!/bin/bash
# round 2, pass ab: hipBLASLt / rocBLAS solution selection for the frozen backbone's GEMMs with PyTorch's TunableOp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ab; mkdir -p $O
export PYTORCH_TUNABLEOP_FILENAME=$GRAFT_REPO_ROOT/$O/tunableop_gfx950.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=8
export PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20
export PYTORCH_TUNABLEOP_VERBOSE=0
t0=$(date +%s)
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 timeout 2400 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_tuning.json.log 2>$O/bench_tuning.err
echo "tuning pass: $(( $(date +%s) - t0 )) s"; ls -la $O/ | head; wc -l $O/tunableop_gfx950*.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_tuned.json.log 2>$O/bench_tuned.err
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_default.json.log 2>$O/bench_default.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ab/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $O/bench_tuning.err
