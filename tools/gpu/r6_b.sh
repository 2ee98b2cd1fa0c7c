#!/bin/bash
# round 6, GPU pass b: in-launch reduce-scatter of the K1 backward (parity incl. held CUs, ABBA A/B vs the finalize launch), K4 suite again
O=gpurun_out/r6b; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1200 python -m pytest tests/test_gpu_cols.py tests/test_gpu_k4.py -x -q 2>&1 | tail -15 | tee $O/pytest_cols_k4.txt
timeout 600 python tools/k1red.py 3500 15272 28000 31616 46648 2>&1 | grep -v amdgpu.ids | tee $O/k1red.txt
K1RED_R=8 timeout 300 python tools/k1red.py 3500 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1red.txt
