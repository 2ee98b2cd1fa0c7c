#!/bin/bash
# round 2, first GPU pass: whole GPU suite, K3 timings, bench lines of the four workloads, DP self-launch smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
timeout 300 python tools/k3bench.py 28000 bf16 > $O/k3bench_28000.txt 2>&1
timeout 300 python tools/kbench.py 28000 bf16 > $O/kbench_28000.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 4 > $O/bench_bart.json.log 2>$O/bench_bart.err
timeout 400 python bench.py --model lora --lora-r 64 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r64.json.log 2>$O/bench_lora64.err
timeout 400 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora_r8.json.log 2>$O/bench_lora8.err
timeout 400 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_video.json.log 2>$O/bench_video.err
timeout 400 python bench.py --gpus 2 --backend gloo --scaling strong --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_strong.json.log 2>$O/bench_dp2.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_lora -o lora -- python bench.py --model lora --lora-r 64 --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_lora.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_video -o video -- python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_video.log 2>&1
find $O -name "*.db" -size +20M -delete
ls -la $O $O/prof_lora $O/prof_video 2>/dev/null | head -50
