#!/bin/bash
# round 2, pass aq: K1 forward with a three-slot weight ring for the 64-row workgroups (M <= 16,384): parity, then A/B against depth 2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aq; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_lowrank.py tests/test_gpu_gates.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest.txt
{
for M in 2048 3500 8000 15272 16384; do
  echo "== ring 3 (default)"; timeout 120 python tools/k1bench.py wr3 $M
  echo "== ring 2";           VLPET_FWD_WR=2 timeout 120 python tools/k1bench.py wr2 $M
done
echo "== f4 ring 3"; timeout 200 python tools/f4bench.py 15000 | grep -v library
echo "== f4 ring 2"; VLPET_FWD_WR=2 timeout 200 python tools/f4bench.py 15000 | grep -v library
} 2>&1 | grep -v amdgpu.ids | tee $O/k1bench_ring.txt
