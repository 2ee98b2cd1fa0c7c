#!/bin/bash
# round 2, pass w: streaming weight-gradient kernel (wgrad_stream_kernel) + parallel finalize: parity, A/B timing, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_k4.py tests/test_gpu_video.py tests/test_gpu_gates.py -m gpu -q -x > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sub.log
tail -5 $O/pytest_sub.log | cut -c1-250
for M in 28000 3500 15272 46648; do
  python tools/kbench.py $M bf16 > $O/kb_${M}_stream.txt 2>&1
  VLPET_WGRAD_STREAM=0 python tools/kbench.py $M bf16 > $O/kb_${M}_old.txt 2>&1
done
VLPET_WGRAD_WGS=384 python tools/kbench.py 28000 bf16 > $O/kb_28000_stream_wgs384.txt 2>&1
VLPET_WGRAD_WGS=512 python tools/kbench.py 28000 bf16 > $O/kb_28000_stream_wgs512.txt 2>&1
VLPET_WGRAD_WGS=192 python tools/kbench.py 28000 bf16 > $O/kb_28000_stream_wgs192.txt 2>&1
VLPET_WGRAD_NSTG=4 python tools/kbench.py 28000 bf16 > $O/kb_28000_stream_nstg4.txt 2>&1
VLPET_WGRAD_NSTG=4 VLPET_WGRAD_WGS=512 python tools/kbench.py 28000 bf16 > $O/kb_28000_stream_nstg4_wgs512.txt 2>&1
grep -H "previous form\|wgrad+fin" $O/kb_*.txt | cut -c1-220
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kb -o kb -- python tools/kbench.py 28000 bf16 > $O/prof_kb.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O/prof_kb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 $f | cut -c1-160
