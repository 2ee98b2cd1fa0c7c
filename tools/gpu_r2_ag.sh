#!/bin/bash
# round 2, pass ag: K1 backward rows kernel with the LDS-DMA issued by the chain that has slack: parity + same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ag; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_video.py tests/test_gpu_gates.py tests/test_host_golden.py -m gpu -q -x > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sub.log
tail -4 $O/pytest_sub.log | cut -c1-250
L=$GRAFT_REPO_ROOT/vl-pet_amd/lib
{
python tools/k1bench.py new 28000 46648 15272 31616 3500
VLPET_LIB=$L/libvlpet_hip_old.so python tools/k1bench.py old 28000 46648 15272 31616 3500
python tools/k1bench.py new 28000
} 2>&1 | grep k1bench | tee $O/k1bench.txt
