#!/bin/bash
# round 4: validation of HEAD -- the whole GPU suite, smoke, the bench line of every config and the emulated per-rank lines
mkdir -p gpurun_out/r4fin2
O=gpurun_out/r4fin2
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
python bench.py > $O/bench_bart.json.log 2> $O/bench_bart.err
python bench.py --model t5 --no-cpu-baseline > $O/bench_t5.json.log 2> $O/bench_t5.err
python bench.py --model lora --no-cpu-baseline > $O/bench_lora.json.log 2> $O/bench_lora.err
python bench.py --model video --no-cpu-baseline > $O/bench_video.json.log 2> $O/bench_video.err
python bench.py --emulate-ranks 8 --no-cpu-baseline > $O/bench_bart_rank1of8_graph.json.log 2> $O/bench_bart_rank1of8.err
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline > $O/bench_t5_rank1of8_graph.json.log 2> $O/bench_t5_rank1of8.err
tail -3 $O/pytest_gpu.txt; tail -2 $O/smoke.txt
