#!/bin/bash
# round 5, GPU pass h: non-temporal loads in the streaming backbone passes (FFN activation, LM-head loss, attention images, Downsample): same-box A/B
O=gpurun_out/r5h; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_act.py tests/test_gpu_attention.py tests/test_gpu_downsample.py tests/test_gpu_loss.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do
for lib in pre new; do
  if [ $lib = pre ]; then export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_pre.so; else unset VLPET_LIB; fi
  timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_bart_${lib}_$rep.json.log 2>&1
  python - $O/bench_bart_${lib}_$rep.json.log $lib <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not l: print(sys.argv[2], "no json"); raise SystemExit
j=json.loads(l[-1]); k=j["kernels"]
print(sys.argv[2], j["value"], j["ms_per_step"], {n:(k[n]["avg_us"]) for n in ("ffn_act_fwd","ffn_act_bwd","attn_fwd","attn_bwd","ce_fwd","ce_bwd") if n in k})
PY
done; done 2>&1 | tee $O/nt_ab.txt
unset VLPET_LIB
