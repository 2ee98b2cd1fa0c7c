#!/bin/bash
# round 5, GPU pass k: T5 -- the residual tail hands dout on as d/dx1 (no copy) and applies the next sublayer's RMS norm in the same pass
O=gpurun_out/r5k; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1200 python -m pytest tests/test_gpu_tail.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_graph.py tests/test_gpu_optim.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.txt
for rep in 1 2; do
for v in head nofuse noboth; do
  unset VLPET_NO_TAIL_NORM_FUSION VLPET_NO_ALIAS_RESIDUAL_GRAD
  if [ $v = nofuse ]; then export VLPET_NO_TAIL_NORM_FUSION=1; fi
  if [ $v = noboth ]; then export VLPET_NO_TAIL_NORM_FUSION=1 VLPET_NO_ALIAS_RESIDUAL_GRAD=1; fi
  VLPET_AB=1 timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_${v}_$rep.json.log 2>&1
  python - $O/bench_t5_${v}_$rep.json.log $v <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not l: print(sys.argv[2], "no json", open(sys.argv[1]).read()[-600:]); raise SystemExit
j=json.loads(l[-1]); k=j["kernels"]
print(sys.argv[2], j["value"], j["ms_per_step"], {n:(k[n]["avg_us"],k[n]["launches"]) for n in ("k5_fwd","k5_bwd","rms_fwd","rms_bwd","k4_fwd") if n in k})
PY
done; done 2>&1 | tee $O/t5_ab.txt
