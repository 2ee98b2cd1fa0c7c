#!/bin/bash
# round 2, pass p: short-sequence attention kernels: parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
O=gpurun_out/r2p
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q > $O/pytest_attn.log 2>&1; echo "rc=$?" >> $O/pytest_attn.log; tail -30 $O/pytest_attn.log | cut -c1-250
