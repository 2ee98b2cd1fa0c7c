#!/bin/bash
# round 6, GPU pass h: localise the memory fault of `bench.py --model lora` (after the pruning commit) + the GPU tests that -x had cut off
O=gpurun_out/r6h; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_loss.py tests/test_gpu_lowrank.py tests/test_gpu_modules.py tests/test_gpu_ng.py tests/test_gpu_optim.py tests/test_gpu_tail.py tests/test_gpu_video.py tests/test_host_golden.py -q 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/pytest.txt
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -X faulthandler bench.py --model lora --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_lora_blocking.log 2>&1
tail -40 $O/bench_lora_blocking.log
