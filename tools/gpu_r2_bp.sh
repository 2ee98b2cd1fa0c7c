#!/bin/bash
# round 2, pass bp: TunableOp entries for the T5 bench's bf16 GEMM shapes (its encoder ran fp32 when the table was measured) and
# for the accumulating dgrad GEMMs of the BART bench, then both benches with the new table
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bp; mkdir -p $O
rm -f gpurun_out/tunableop_gfx950_new.csv
timeout 1500 python bench.py --model t5 --steps 4 --warmup 2 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_t5.log 2>&1; tail -1 $O/tune_t5.log | cut -c1-200
timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --kernel-table off --gemm-table tune > $O/tune_bart.log 2>&1; tail -1 $O/tune_bart.log | cut -c1-200
ls -la gpurun_out/tunableop_gfx950_new.csv*; wc -l gpurun_out/tunableop_gfx950_new.csv*
cp gpurun_out/tunableop_gfx950_new.csv $O/tunableop_gfx950_new.csv
cp gpurun_out/tunableop_gfx950_new.csv vl-pet_amd/tuning/tunableop_gfx950.csv
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_t5_newtable.json.log 2>$O/t.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_bart_newtable.json.log 2>$O/b.err
git checkout vl-pet_amd/tuning/tunableop_gfx950.csv 2>/dev/null || true
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bp/bench_*.json.log")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
PY
