#!/bin/bash
# round 2, pass be: K4's LayerNorm backward on K5's backward kernel (vlpet_layernorm_bwd_xhat) -- parity, then op sources again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2be; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_k4.py tests/test_gpu_tail.py tests/test_gpu_lowrank.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_video.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_subset.txt; tail -3 $O/pytest_subset.txt
timeout 500 python tools/opcount.py vqa 2>&1 | grep -v amdgpu.ids > $O/opcount_vqa.txt; head -30 $O/opcount_vqa.txt
