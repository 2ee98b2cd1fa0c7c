#!/bin/bash
# round 2, pass am: f4 (low-rank visual projector on the rectangular K1 kernels) -- first run of the new kernels:
# its parity tests, then the suites of the kernels it shares code with (K1, K5, K4, gates, modules), then K1 timings
# (the LR switches are template parameters; the K1 numbers must not move)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2am; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lowrank.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_lowrank.txt
echo "lowrank rc=$?" >> $O/pytest_lowrank.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tail.py tests/test_gpu_k4.py tests/test_gpu_gates.py tests/test_gpu_modules.py tests/test_gpu_video.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -25 > $O/pytest_shared.txt
timeout 300 python tools/k1bench.py am 3500 28000 2>&1 | grep -v amdgpu.ids > $O/k1bench.txt
timeout 300 python tools/k5bench.py 28000 2>&1 | grep -v amdgpu.ids > $O/k5bench.txt
tail -5 $O/pytest_lowrank.txt; tail -3 $O/pytest_shared.txt; cat $O/k1bench.txt $O/k5bench.txt
