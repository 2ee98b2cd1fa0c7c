#!/bin/bash
# round 5, GPU pass d: whole GPU suite + the four bench configs at HEAD
O=gpurun_out/r5d; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_bart.json.log 2>&1; tail -c 600 $O/bench_bart.json.log | head -c 300; echo
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 > $O/bench_t5.json.log 2>&1; tail -c 300 $O/bench_t5.json.log | head -c 200; echo
timeout 600 python bench.py --model lora --steps 12 --warmup 4 > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model video --steps 8 --warmup 3 > $O/bench_video.json.log 2>&1
