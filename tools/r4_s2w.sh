#!/bin/bash
mkdir -p gpurun_out/r4ai
O=gpurun_out/r4ai
timeout 1500 python -m pytest tests/test_host_golden.py tests/test_gpu_optim.py tests/test_gpu_graph.py tests/test_gpu_modules.py tests/test_gpu_tail.py tests/test_gpu_dp.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o kt -- python bench.py --model lora --no-cpu-baseline --kernel-table off --steps 12 > $O/bench_lora_prof.json.log 2> $O/err.txt
find $O/prof -name "kt_kernel_stats.csv" -exec cp {} $O/kernel_stats_lora.csv \;
rm -rf $O/prof
python bench.py --model lora --emulate-ranks 8 --no-cpu-baseline > $O/bench_lora_rank1of8_graph.json.log 2>> $O/err.txt
python bench.py --model lora --no-cpu-baseline > $O/bench_lora.json.log 2>> $O/err.txt
