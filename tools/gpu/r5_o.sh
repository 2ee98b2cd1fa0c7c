#!/bin/bash
# round 5, GPU pass o: whole GPU suite with the deferred finalize / from-the-output backward / reordered pass prologues, then A/B of
# the deferred finalize in the step (full batch and emulated rank 1 of 8) and K1 through the C ABI
O=gpurun_out/r5o; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_gpu.txt
for r in 96 192; do
  SZ="3500 8232 28000"; [ $r = 192 ] && SZ="2128 18250"
  K1BENCH_R=$r python tools/k1bench.py head $SZ 2>&1 | grep -v amdgpu.ids | tee -a $O/k1bench.txt
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
python - <<'P' | tee gpurun_out/r5o/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5o/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]; ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "wgrad", k["k1_bwd_wgrad"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
    if not ok: print(f, "NO JSON LINE"); print(open(f).read()[-1500:])
P
