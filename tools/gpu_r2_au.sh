#!/bin/bash
# round 2, pass au: where does the weight-gradient plan cross over from <= 256 workgroups to ~512 (cold inputs, no row cap)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2au; mkdir -p $O
{
for M in 24000 28000 31616 33200 36000 40000 46648 56000; do
  for W in 256 512; do
    K1BENCH_COLD=1 VLPET_WGRAD_MAXROWS=0 VLPET_WGRAD_WGS=$W timeout 200 python tools/k1bench.py w$W $M | sed -E 's/fwd\+save +[0-9.]+ us +bwd rows +[0-9.]+ us +//'
  done
done
} 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_crossover.txt
