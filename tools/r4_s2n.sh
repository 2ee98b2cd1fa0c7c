#!/bin/bash
mkdir -p gpurun_out/r4z
O=gpurun_out/r4z
timeout 1500 python -m pytest tests/test_host_golden.py tests/test_gpu_optim.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_tail.py tests/test_gpu_modules.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.txt
python bench.py --model lora --no-cpu-baseline > $O/bench_lora.json.log 2> $O/bench_lora.err
python bench.py --no-cpu-baseline > $O/bench_bart.json.log 2> $O/bench_bart.err
python bench.py --model lora --emulate-ranks 8 --no-cpu-baseline > $O/bench_lora_rank1of8_graph.json.log 2> $O/bench_lora_r8.err
tail -2 $O/bench_lora.err
