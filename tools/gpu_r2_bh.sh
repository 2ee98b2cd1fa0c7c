#!/bin/bash
# round 2, pass bh: FFN activation pass with one-exp / one-rcp transcendental forms and two groups per thread in flight --
# parity, then same-box A/B against the previous binary (tools/_lib_old.so, not committed)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bh; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_act.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest_act.txt
echo "== old" | tee $O/actbench.txt; VLPET_LIB=$GRAFT_REPO_ROOT/tools/_lib_old.so timeout 300 python tools/actbench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/actbench.txt
echo "== new" | tee -a $O/actbench.txt; timeout 300 python tools/actbench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/actbench.txt
