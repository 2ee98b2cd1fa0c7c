#!/bin/bash
# round 5, GPU pass w: PMC traffic of the K1 backward as it runs now (from the forward's output; r = 192 on pet_cols6y.hip) and of K5 (branch-free
# row kernels), GPU suite, rocprofv3 kernel statistics of the BART / T5 bench commands
O=gpurun_out/r5w; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
bash tools/pmc_traffic4.sh $O/pmc "k1bwd 28000" "k1bwd 46648" "k1bwd 15272" "k1bwd 31616" "k1bwd 18250 192" "k1bwd 28000 192" "k5fwd 28000" "k5bwd 28000"
cp $O/pmc/summary.txt $O/pmc_summary.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o kt -- python bench.py --steps 8 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_bart_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t5 -o kt -- python bench.py --model t5 --steps 8 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_t5_prof.log 2>&1
for m in bart t5; do f=$(find $O/prof_$m -name "kt_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$m.csv; rm -rf $O/prof_$m; done
ls -la $O
