#!/bin/bash
# round 5, GPU pass c: K4 tiled GEMM -- ring forms with / without spread requests, phase stamps
O=gpurun_out/r5c; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 600 python -m pytest tests/test_gpu_k4.py -m gpu -q -x -k "gemm" 2>&1 | tail -4 | tee $O/pytest_k4.txt
timeout 600 python tools/k4bench.py r5c 7200 10800 18700 29988 2>&1 | grep -v amdgpu.ids | tee $O/k4bench.txt
