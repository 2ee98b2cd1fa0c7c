#!/bin/bash
# round 4, second session: T5's biased attention on the short-sequence kernels -- parity, then same-box A/B of the T5 step
mkdir -p gpurun_out/r4v
O=gpurun_out/r4v
timeout 1200 python -m pytest tests/test_gpu_attention.py tests/test_host_golden.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest.txt
python bench.py --model t5 --no-cpu-baseline > $O/bench_t5.json.log 2> $O/bench_t5.err
VLPET_AB=1 VLPET_EAGER_ATTENTION=1 python bench.py --model t5 --no-cpu-baseline > $O/bench_t5_sdpa.json.log 2> $O/bench_t5_sdpa.err
python bench.py --model t5 --emulate-ranks 8 --no-cpu-baseline > $O/bench_t5_rank1of8_graph.json.log 2> $O/bench_t5_rank1of8.err
tail -3 $O/bench_t5.err
