#!/bin/bash
# round 5, GPU pass e: PMC traffic of the K1 backward at the four task sizes of configs[1] (+ T5 r = 192, K5, K4), rocprofv3 kernel
# statistics of the BART / T5 bench commands, the K1 / K5 parity suites with the diagnosis library present
O=gpurun_out/r5e; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tail.py tests/test_gpu_cols.py -m gpu -q -s 2>&1 | grep -E "element-wise|passed|failed" | tail -30 | tee $O/pytest.txt
bash tools/pmc_traffic4.sh $O/pmc "k1bwd 28000" "k1bwd 46648" "k1bwd 15272" "k1bwd 31616" "k1bwd 18250 192" "k1bwd 28000 192" "k1fwd 28000" "k5fwd 28000" "k5bwd 28000" "k4fwd 18700"
cp $O/pmc/summary.txt $O/pmc_summary.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o kt -- python bench.py --steps 8 --warmup 4 --kernel-table off > $O/bench_bart_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t5 -o kt -- python bench.py --model t5 --steps 8 --warmup 4 --kernel-table off > $O/bench_t5_prof.log 2>&1
for m in bart t5; do f=$(find $O/prof_$m -name "kt_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$m.csv; rm -rf $O/prof_$m; done
ls -la $O
