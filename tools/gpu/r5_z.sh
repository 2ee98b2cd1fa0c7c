#!/bin/bash
# round 5, GPU pass z: DP tests incl. the captured-collectives graph mode on a one-rank RCCL communicator; the alternative-form child tests
O=gpurun_out/r5z; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_graph.py tests/test_gpu_ng.py -m gpu -q 2>&1 | tail -15 | tee $O/pytest.txt
