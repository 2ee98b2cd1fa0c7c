#!/bin/bash
# round 4, second session, run 5: graph-mode trainer tests + emulated-rank bench lines in graph mode
mkdir -p gpurun_out/r4q
O=gpurun_out/r4q
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_tail.py tests/test_gpu_act.py tests/test_gpu_attention.py tests/test_gpu_optim.py -m gpu -q -x 2>&1 | tail -25 | tee $O/pytest_graph.txt
python bench.py --emulate-ranks 8 --no-cpu-baseline > $O/bench_bart_rank1of8_graph.json.log 2> $O/bench_bart_rank1of8_graph.err
python bench.py --emulate-ranks 8 --no-cpu-baseline --graph off > $O/bench_bart_rank1of8_eager.json.log 2> $O/bench_bart_rank1of8_eager.err
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline > $O/bench_t5_rank1of8_graph.json.log 2> $O/bench_t5_rank1of8_graph.err
python bench.py --model lora --emulate-ranks 8 --no-cpu-baseline > $O/bench_lora_rank1of8_graph.json.log 2> $O/bench_lora_rank1of8_graph.err
python bench.py --emulate-ranks 2 --no-cpu-baseline > $O/bench_bart_rank1of2_graph.json.log 2> $O/bench_bart_rank1of2_graph.err
python bench.py --emulate-ranks 4 --no-cpu-baseline > $O/bench_bart_rank1of4_graph.json.log 2> $O/bench_bart_rank1of4_graph.err
python bench.py --graph on --no-cpu-baseline > $O/bench_bart_graph_on.json.log 2> $O/bench_bart_graph_on.err
tail -3 $O/*.err
