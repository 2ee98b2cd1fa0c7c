#!/bin/bash
mkdir -p gpurun_out/r4ac
O=gpurun_out/r4ac
timeout 900 python bench.py --gpus 2 --backend gloo --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_dp2_gloo_graph.json.log 2> $O/bench_dp2_gloo_graph.err
echo "rc=$?" >> $O/bench_dp2_gloo_graph.err
timeout 900 python bench.py --gpus 2 --backend gloo --steps 8 --warmup 2 --no-cpu-baseline --graph off > $O/bench_dp2_gloo_eager.json.log 2> $O/bench_dp2_gloo_eager.err
echo "rc=$?" >> $O/bench_dp2_gloo_eager.err
tail -4 $O/*.err
