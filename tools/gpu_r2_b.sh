#!/bin/bash
# round 2, second GPU pass: whole GPU suite with the two-pass K1 backward, A/B timings, bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
for M in 28000 46648 15272 3500 512; do timeout 300 python tools/kbench.py $M bf16 > $O/kbench_$M.txt 2>&1; done
timeout 300 python tools/kbench.py 28000 fp32 > $O/kbench_28000_fp32.txt 2>&1
grep -h "two-pass\|previous form" $O/kbench_*.txt
timeout 900 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>$O/bench_bart.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>$O/bench_lora8.err
timeout 400 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_video.json.log 2>$O/bench_video.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lora -o lora -- python bench.py --model lora --lora-r 64 --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_lora.log 2>&1
find $O -name "*_kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
ls -la $O $O/prof_bart $O/prof_lora 2>/dev/null | head -50
