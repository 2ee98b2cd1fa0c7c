#!/bin/bash
# round 2, pass bk: attention backward with the phase-K register diet (174 registers; three waves per SIMD spill 4) --
# parity, then timings at two and three waves per SIMD (VLPET_ATTN_OCC=3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bk; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attention.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/pytest_attn.txt
VLPET_ATTN_OCC=3 timeout 600 python -m pytest tests/test_gpu_attention.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/pytest_attn.txt
echo "== occ 2" | tee $O/attn.txt; timeout 300 python tools/attnbench2.py 2>&1 | grep -v amdgpu.ids | grep enc | tee -a $O/attn.txt
echo "== occ 3" | tee -a $O/attn.txt; VLPET_ATTN_OCC=3 timeout 300 python tools/attnbench2.py 2>&1 | grep -v amdgpu.ids | grep enc | tee -a $O/attn.txt
