#!/bin/bash
# round 2, pass as: weight-gradient kernel under COLD inputs (the in-step condition): workgroup count and ring depth were tuned warm
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2as; mkdir -p $O
{
for cfg in "256 3" "384 3" "512 3" "768 3" "512 4" "256 4"; do
  set -- $cfg
  echo "== VLPET_WGRAD_WGS=$1 VLPET_WGRAD_NSTG=$2"
  for M in 28000 46648; do K1BENCH_COLD=1 VLPET_WGRAD_WGS=$1 VLPET_WGRAD_NSTG=$2 timeout 200 python tools/k1bench.py w$1s$2 $M; done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_cold.txt
