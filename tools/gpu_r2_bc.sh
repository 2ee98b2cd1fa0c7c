#!/bin/bash
# round 2, pass bc: LayerNorm parameter gradients of K5 in one launch into the flat gradient buffer, frozen LayerNorm copies
# cached -- tail / module / whole-model / DP parity, then rocprofv3 kernel statistics of the default bench + bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bc; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tail.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_dp.py tests/test_gpu_optim.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_subset.txt; tail -3 $O/pytest_subset.txt
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bart -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bart_kernel_stats.csv
rm -rf $O/prof
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench.json.log 2> $O/b.err
tail -c 600 $O/bench.json.log | head -c 300
