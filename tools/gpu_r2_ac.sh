#!/bin/bash
# round 2, pass ac: extend the TunableOp table to the other workloads (t5, lora r = 64 / 8, video), then every bench line with it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ac; mkdir -p $O
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=8
export PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=20
t0=$(date +%s)
timeout 1500 python bench.py --model t5 --steps 4 --warmup 4 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_t5.log 2>&1
timeout 1500 python bench.py --model lora --lora-r 64 --steps 4 --warmup 4 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_lora.log 2>&1
timeout 1500 python bench.py --model video --steps 4 --warmup 4 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_video.log 2>&1
echo "tuning passes: $(( $(date +%s) - t0 )) s"; wc -l gpurun_out/tunableop_gfx950_new.csv
cp gpurun_out/tunableop_gfx950_new.csv vl-pet_amd/tuning/tunableop_gfx950.csv
cp gpurun_out/tunableop_gfx950_new.csv $O/tunableop_gfx950.csv
timeout 900 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>$O/bench_bart.err
for m in t5 video; do
  timeout 400 python bench.py --model $m --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_$m.json.log 2>$O/bench_$m.err
  timeout 400 python bench.py --model $m --steps 12 --warmup 4 --no-cpu-baseline --gemm-table off > $O/bench_${m}_untuned.json.log 2>>$O/bench_$m.err
done
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline --gemm-table off > $O/bench_lora_r64_untuned.json.log 2>>$O/bench_lora.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ac/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j.get("backbone_gemm_table"))
    except Exception as e: print(f, "ERR", e)
PY
