#!/bin/bash
# round 5, GPU pass q: whole GPU suite (process-wide finalize queue), A/B of the three round-5 changes to pass 1 at six tiles (variant
# libraries built from pet_dz6.hip's switches), the deferred finalize in the step for real
O=gpurun_out/r5q; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
for rep in 1 2; do
for v in "" _dz6_late _dz6_nopin _dz6_episerial _dz6_r4like; do
  echo "== lib$v" | tee -a $O/k1bench_dz6.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip$v.so K1BENCH_R=192 python tools/k1bench.py "head$v" 2128 18250 28000 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench_dz6.txt
done
done
for rep in 1 2; do
  for m in bart t5; do
    extra=""; [ $m = t5 ] && extra="--model t5"
    timeout 600 python bench.py $extra --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_${m}_defer_$rep.json.log 2>&1
    VLPET_AB=1 VLPET_NO_DEFER_FINALIZE=1 timeout 600 python bench.py $extra --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_${m}_nodefer_$rep.json.log 2>&1
    timeout 600 python bench.py $extra --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_${m}_rank1of8_defer_$rep.json.log 2>&1
    VLPET_AB=1 VLPET_NO_DEFER_FINALIZE=1 timeout 600 python bench.py $extra --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_${m}_rank1of8_nodefer_$rep.json.log 2>&1
  done
done
python - <<'P' | tee gpurun_out/r5q/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5q/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "wgrad", k["k1_bwd_wgrad"]["avg_us"], "k5_fwd", k["k5_fwd"]["avg_us"], "k5_bwd", k["k5_bwd"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
    if not ok: print(f, "NO JSON LINE"); print(open(f).read()[-1500:])
P
