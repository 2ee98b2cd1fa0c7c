#!/bin/bash
# round 6, GPU pass u: does the committed TunableOp table still cover every library-GEMM shape of the bench's steps (N = 1 and the per-rank
# batches of 2 / 4 / 8 ranks)?  --gemm-table tune appends what is missing to gpurun_out/tunableop_gfx950_new.csv (it starts from the committed table)
O=gpurun_out/r6u; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
rm -f gpurun_out/tunableop_gfx950_new.csv
for r in 1 2 4 8; do
  timeout 1500 python bench.py --gemm-table tune --graph off --steps 4 --warmup 1 --kernel-table off --no-cpu-baseline $( [ $r != 1 ] && echo --emulate-ranks $r ) > $O/tune_r$r.log 2>&1
  echo "ranks $r rc=$? lines $(wc -l < gpurun_out/tunableop_gfx950_new.csv)" | tee -a $O/summary.txt
done
cp gpurun_out/tunableop_gfx950_new.csv $O/
diff <(sort vl-pet_amd/tuning/tunableop_gfx950.csv) <(sort gpurun_out/tunableop_gfx950_new.csv) | head -80 | tee -a $O/summary.txt
