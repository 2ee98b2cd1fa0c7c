#!/bin/bash
# round 5, GPU pass i: cache policy of pass 1's row loads (dy, x2 are read again by pass 2 right after): nt (aux 2, HEAD) vs default (aux 0)
O=gpurun_out/r5i; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for lib in head p1aux0; do
  if [ $lib = p1aux0 ]; then export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_p1aux0.so; else unset VLPET_LIB; fi
  K1BENCH_COLD=1 python tools/k1bench.py $lib 15272 28000 31616 46648 2>&1 | grep k1bench
  K1BENCH_COLD=1 K1BENCH_R=192 python tools/k1bench.py $lib-r192 18250 2>&1 | grep k1bench
done; done 2>&1 | tee $O/k1bench_cold.txt
for rep in 1 2; do
for lib in head p1aux0; do
  if [ $lib = p1aux0 ]; then export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_p1aux0.so; else unset VLPET_LIB; fi
  timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_bart_${lib}_$rep.json.log 2>&1
  python - $O/bench_bart_${lib}_$rep.json.log $lib <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not l: print(sys.argv[2], "no json"); raise SystemExit
j=json.loads(l[-1]); k=j["kernels"]
print(sys.argv[2], j["value"], j["ms_per_step"], {n:(k[n]["avg_us"]) for n in ("k1_bwd_rows","k1_bwd_wgrad","k1_bwd_fin","k1_bwd_op","k1_fwd","k5_bwd") if n in k})
PY
done; done 2>&1 | tee $O/instep_ab.txt
unset VLPET_LIB
