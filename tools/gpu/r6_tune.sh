#!/bin/bash
# round 6, GPU pass tune: the fused cross-attention key projection brought new library-GEMM shapes ([M, 768] x [768, n_layers * 768] and its
# dgrad): measure them into the TunableOp table (bench.py --gemm-table tune starts from the committed table and appends what is missing)
O=gpurun_out/r6tune; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
rm -f gpurun_out/tunableop_gfx950_new.csv
for cfg in "" "--emulate-ranks 2" "--emulate-ranks 4" "--emulate-ranks 8" "--model t5" "--model t5 --emulate-ranks 2" "--model t5 --emulate-ranks 4" "--model t5 --emulate-ranks 8" "--model video" "--model lora"; do
  timeout 1500 python bench.py --gemm-table tune --graph off --steps 4 --warmup 1 --kernel-table off --no-cpu-baseline $cfg > "$O/tune_$(echo $cfg | tr -d ' -').log" 2>&1
  echo "[$cfg] rc=$? lines $(wc -l < gpurun_out/tunableop_gfx950_new.csv)" | tee -a $O/summary.txt
done
cp gpurun_out/tunableop_gfx950_new.csv $O/
diff <(sort vl-pet_amd/tuning/tunableop_gfx950.csv) <(sort gpurun_out/tunableop_gfx950_new.csv) | grep "^>" | tee -a $O/summary.txt | head -60
