#!/bin/bash
# round 5, GPU pass x: K3 backward at r <= 8 as the streaming row kernel -- parity, C ABI A/B (debug build, VLPET_LORA8_BWD=0 = two-pass MFMA form), LoRA step
O=gpurun_out/r5x; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_ng.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_graph.py -m gpu -q 2>&1 | tail -12 | tee $O/pytest.txt
for rep in 1 2; do
  for v in 1 0; do
    for M in 2500 10000 16640 28000; do
      echo "== debug build, VLPET_LORA8_BWD=$v M=$M" | tee -a $O/k3bench.txt
      VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_LORA8_BWD=$v python tools/k3bench.py $M 2>&1 | grep -v amdgpu.ids | grep -E "^r=   8" | cut -c1-260 | tee -a $O/k3bench.txt
    done
  done
  timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora8_$rep.json.log 2>&1
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_LORA8_BWD=0 timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora8_mfma_bwd_$rep.json.log 2>&1
done
python - <<'P' | tee gpurun_out/r5x/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5x/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], {n: k[n]["avg_us"] for n in ("k3_fwd", "k3_bwd") if n in k})
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1200:])
P
