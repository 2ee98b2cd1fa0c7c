#!/bin/bash
# round 2, pass u: attention parity + microbench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
O=gpurun_out/r2u
timeout 600 python -m pytest tests/test_gpu_attention.py -m gpu -q > $O/pytest_attn.log 2>&1; echo "rc=$?" >> $O/pytest_attn.log; tail -2 $O/pytest_attn.log | cut -c1-250
timeout 300 python tools/attnbench.py 2>&1 | tee $O/attnbench.txt
