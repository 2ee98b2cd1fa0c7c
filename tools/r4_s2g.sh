#!/bin/bash
# GEMM solution table (TunableOp) for the strong-scaled per-rank shapes: measure the missing shapes of the emulated 8 / 4 / 2-rank batches
mkdir -p gpurun_out/r4s
O=gpurun_out/r4s
rm -f gpurun_out/tunableop_gfx950_new.csv
for spec in "bart 8" "bart 4" "bart 2" "lora 8" "t5 8" "video 8"; do
  set -- $spec
  timeout 1500 python bench.py --model $1 --emulate-ranks $2 --no-cpu-baseline --gemm-table tune --kernel-table off --steps 8 > $O/tune_$1_$2.json.log 2> $O/tune_$1_$2.err
  wc -l gpurun_out/tunableop_gfx950_new.csv | tee -a $O/log.txt
done
cp gpurun_out/tunableop_gfx950_new.csv $O/tunableop_gfx950_new.csv
