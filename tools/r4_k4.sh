#!/bin/bash
mkdir -p gpurun_out/r4i
O=gpurun_out/r4i
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py tests/test_gpu_tail.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_video.py -m gpu -q 2>&1 | tail -5 | tee $O/log.txt
python tools/k4bench.py r4 2592 18000 18700 29988 2>&1 | grep k4bench | tee $O/k4bench.txt
