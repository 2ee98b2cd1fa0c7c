#!/bin/bash
# round 4, second session, run 3: the whole GPU suite + the bench lines of every config at HEAD
mkdir -p gpurun_out/r4o
O=gpurun_out/r4o
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
python bench.py > $O/bench_bart.json.log 2> $O/bench_bart.err
python bench.py --model lora --no-cpu-baseline > $O/bench_lora.json.log 2> $O/bench_lora.err
python bench.py --model t5 --no-cpu-baseline > $O/bench_t5.json.log 2> $O/bench_t5.err
python bench.py --model video --no-cpu-baseline > $O/bench_video.json.log 2> $O/bench_video.err
python bench.py --emulate-ranks 8 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2> $O/bench_bart_rank1of8.err
python bench.py --model lora --emulate-ranks 8 --no-cpu-baseline > $O/bench_lora_rank1of8.json.log 2> $O/bench_lora_rank1of8.err
