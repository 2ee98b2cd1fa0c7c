#!/bin/bash
# round 5, GPU pass ze: pass 2 at r <= 96, second form (pet_colsy.hip) -- parity, C ABI A/B (debug build, VLPET_COLSY=0 = pet_cols.hip), BART step
O=gpurun_out/r5ze; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_video.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_gates.py -m gpu -q 2>&1 | tail -12 | tee $O/pytest.txt
for rep in 1 2; do
  for v in 1 0; do
    echo "== debug build, VLPET_COLSY=$v" | tee -a $O/k1bench.txt
    VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_COLSY=$v python tools/k1bench.py "colsy=$v" 3500 15272 28000 31616 46648 2>&1 | grep -v amdgpu.ids | sed 's/| previous split.*| default://' | tee -a $O/k1bench.txt
  done
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_COLSY=1 K1BENCH_COLD=1 python tools/k1bench.py "colsy=1" 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_COLSY=0 K1BENCH_COLD=1 python tools/k1bench.py "colsy=0" 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench.txt
  timeout 600 python bench.py --steps 16 --warmup 5 --no-cpu-baseline > $O/bench_bart_$rep.json.log 2>&1
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_COLSY=0 timeout 600 python bench.py --steps 16 --warmup 5 --no-cpu-baseline > $O/bench_bart_cols_$rep.json.log 2>&1
done
python - <<'P' | tee gpurun_out/r5ze/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5ze/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "wgrad", k["k1_bwd_wgrad"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1200:])
P
