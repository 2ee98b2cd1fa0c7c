#!/bin/bash
O=gpurun_out/r6j; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -X faulthandler bench.py --model lora --graph off --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_lora_eager_blocking.log 2>&1
tail -60 $O/bench_lora_eager_blocking.log | cut -c1-300
timeout 300 python -m pytest tests/test_gpu_k4.py -q -x -k give_up 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-300
