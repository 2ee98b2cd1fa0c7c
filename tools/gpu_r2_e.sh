#!/bin/bash
# round 2, pass e: K4 column-split kernel (parity + timing A/B), K3 training form timing, PMC traffic of the K1 backward, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
timeout 900 python -m pytest tests/test_gpu_k4.py tests/test_gpu_video.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_host_golden.py -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 300 python tools/k3bench.py 28000 bf16 > $O/k3bench_28000.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_K4_WAVES4=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_k4old.json.log 2>$O/bench_bart_k4old.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_video.json.log 2>$O/bench_video.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O/pmc
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc -o bwd$i -- python tools/kfwd_only.py 28000 bwd > $O/pmc/log_bwd$i.txt 2>&1
done
python tools/pmc_summary.py $O/pmc "" > $O/pmc_summary_k1_bwd.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
ls -la $O | head -40; cat $O/pmc_summary_k1_bwd.txt | head -60
