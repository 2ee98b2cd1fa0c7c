#!/bin/bash
# round 2, third GPU pass: pass-2 workgroup shapes A/B, r = 192 fused path, K1 parity subset
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_video.py tests/test_gpu_optim.py -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for cfg in 0 1 2; do
  for M in 28000 46648; do VLPET_BWD3_CFG=$cfg timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_cfg$cfg.txt 2>&1; done
done
VLPET_BWD3_CFG=2 timeout 300 python tools/kbench.py 3500 bf16 > $O/kbench_3500_cfg2.txt 2>&1
VLPET_BWD3_CFG=2 timeout 300 python tools/kbench.py 28000 fp32 > $O/kbench_28000_fp32_cfg2.txt 2>&1
timeout 300 python tools/kbench.py 16800 bf16 192 > $O/kbench_16800_r192.txt 2>&1
grep -H "two-pass\|previous form" $O/kbench_*.txt
timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_split.json.log 2>$O/bench_t5_split.err
VLPET_FUSED_WIDE=1 timeout 400 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_fused.json.log 2>$O/bench_t5_fused.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>$O/bench_bart.err
ls -la $O | head -40
