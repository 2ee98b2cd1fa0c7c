#!/bin/bash
# round 5, GPU pass r: K3 streaming forward (r <= 8) with the generator-only mask form vs the any-source form; pass 1 at six tiles without its
# act' loads (lower bound of the epilogue); GPU suite
O=gpurun_out/r5r; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
for rep in 1 2; do
  for v in "" _lora8_general; do
    echo "== lib$v" | tee -a $O/k3bench.txt
    VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so python tools/k3bench.py "r8$v" 8 2500 10000 16640 28000 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee -a $O/k3bench.txt
  done
  for v in "" _dz6_noload; do
    echo "== lib$v" | tee -a $O/k1bench_dz6.txt
    VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so K1BENCH_R=192 python tools/k1bench.py "head$v" 18250 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench_dz6.txt
  done
  timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora8_$rep.json.log 2>&1
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_lora8_general.so timeout 600 python bench.py --model lora --lora-r 8 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora8_general_$rep.json.log 2>&1
done
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora64.json.log 2>&1
python - <<'P' | tee gpurun_out/r5r/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5r/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], {n: k[n]["avg_us"] for n in ("k3_fwd", "k3_bwd", "k5_fwd", "k5_bwd") if n in k})
    if not ok: print(f, "NO JSON LINE"); print(open(f).read()[-1500:])
P
