#!/bin/bash
# round 2, pass d: full GPU suite with the new defaults; weight-gradient kernel variants A/B; bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
for M in 28000 46648; do
  timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_base.txt 2>&1
  VLPET_WGRAD_TR=1 timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_tr.txt 2>&1
  VLPET_WGRAD_TR=1 VLPET_WGRAD_WGS=512 timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_tr512.txt 2>&1
  VLPET_WGRAD_TR=1 VLPET_WGRAD_WGS=1024 timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_tr1024.txt 2>&1
  VLPET_WGRAD_WGS=1024 timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_base1024.txt 2>&1
done
grep -H "bwd wgrad+fin\|previous form\|bwd rows krn" $O/kbench_*.txt
timeout 900 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>$O/bench_bart.err
VLPET_WGRAD_TR=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart_tr.json.log 2>$O/bench_bart_tr.err
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>$O/bench_t5.err
ls -la $O | head -40
