#!/bin/bash
# round 5, GPU pass s: pass 1 at six tiles with the act' rows through LDS (vs the serial epilogue), parity + C ABI + T5 step
O=gpurun_out/r5s; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_video.py tests/test_gpu_modules.py -m gpu -q 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2; do
  for v in "" _dz6_serial; do
    echo "== lib$v" | tee -a $O/k1bench_dz6.txt
    VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so K1BENCH_R=192 python tools/k1bench.py "head$v" 2128 18250 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench_dz6.txt
  done
  K1BENCH_COLD=1 K1BENCH_R=192 python tools/k1bench.py head 18250 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench_dz6.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dz6_serial.so K1BENCH_COLD=1 K1BENCH_R=192 python tools/k1bench.py head_dz6_serial 18250 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench_dz6.txt
  timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_$rep.json.log 2>&1
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dz6_serial.so timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_serial_$rep.json.log 2>&1
done
python - <<'P' | tee gpurun_out/r5s/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5s/bench_*.json.log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "wgrad", k["k1_bwd_wgrad"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
P
