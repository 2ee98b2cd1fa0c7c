#!/bin/bash
# round 2, pass f: full GPU suite (K3 training form with the packed mask), K3 timings, LoRA bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
O=gpurun_out/r2f
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 400 python tools/k3bench.py 28000 bf16 > $O/k3bench_28000.txt 2>&1
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>$O/bench_lora8.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lora -o lora -- python bench.py --model lora --lora-r 64 --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_lora.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_video -o video -- python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_video.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
ls -la $O | head; grep -v amdgpu $O/k3bench_28000.txt
