#!/bin/bash
# round 2, pass ao: rocprofv3 kernel statistics of the f4 bench (csv), then the whole GPU suite, smoke() and the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ao; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o f4 -- python $GRAFT_REPO_ROOT/tools/f4bench.py 18700 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/f4_kernel_stats.csv && head -12 $O/f4_kernel_stats.csv | cut -c1-220
find $O/prof -name "*trace.csv" -size +4M -delete
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench.err; tail -c 1500 $O/bench_default.json.log | cut -c1-1500
