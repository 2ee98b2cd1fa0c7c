#!/bin/bash
# round 2, pass o: kernel profile of the step after the FFN-activation and LM-loss fusions
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2o
O=gpurun_out/r2o
timeout 600 python -m pytest tests/test_gpu_loss.py -m gpu -q > $O/pytest_loss.log 2>&1; echo "rc=$?" >> $O/pytest_loss.log; tail -3 $O/pytest_loss.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
tail -2 $O/prof_bart.log | cut -c1-300
