#!/bin/bash
# round 6, GPU pass t: the N > 1 code path of bench.py at HEAD (replay by default, eager region, settling all-reduce, input buffers):
# two gloo ranks sharing the one GPU, launched the way the driver launches N ranks; and the launcher-less form
O=gpurun_out/r6t; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo > $O/bench_dp2_gloo.json.log 2>&1
echo "rc=$?" >> $O/bench_dp2_gloo.json.log
timeout 900 python bench.py --gpus 2 --steps 8 --warmup 2 --backend gloo --scaling weak --batch 200 > $O/bench_dp2_gloo_weak.json.log 2>&1
echo "rc=$?" >> $O/bench_dp2_gloo_weak.json.log
grep -h "^{\|rc=\|Error\|error" $O/*.log | cut -c1-600
