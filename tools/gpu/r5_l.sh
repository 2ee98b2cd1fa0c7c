#!/bin/bash
# round 5, GPU pass l: whole GPU suite + smoke + the four bench configs + rocprofv3 kernel statistics at HEAD
O=gpurun_out/r5l; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_video.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o kt -- python bench.py --steps 8 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_bart_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t5 -o kt -- python bench.py --model t5 --steps 8 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_t5_prof.log 2>&1
for m in bart t5; do f=$(find $O/prof_$m -name "kt_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$m.csv; rm -rf $O/prof_$m; done
