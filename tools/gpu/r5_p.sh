#!/bin/bash
# round 5, GPU pass p: the DP test that failed in pass o (traceback), the rest of the GPU suite without -x, K5 / K1 through the C ABI with
# the branch-free row kernels and the batched dz6 epilogue
O=gpurun_out/r5p; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -q 2>&1 | tail -60 > $O/pytest_dp.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python tools/k5abi.py 1000 2000 3500 6000 10000 16640 28000 46648 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee $O/k5abi.txt
for r in 96 192; do
  SZ="3500 8232 28000"; [ $r = 192 ] && SZ="2128 18250 28000"
  K1BENCH_R=$r python tools/k1bench.py head $SZ 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench.txt
done
timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_bart.json.log 2>&1
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
python - <<'P' | tee gpurun_out/r5p/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5p/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "wgrad", k["k1_bwd_wgrad"]["avg_us"], "k5_fwd", k["k5_fwd"]["avg_us"], "k5_bwd", k["k5_bwd"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
    if not ok: print(f, "NO JSON LINE"); print(open(f).read()[-1500:])
P
