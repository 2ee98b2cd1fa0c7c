#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_k4.txt
timeout 600 python tools/k4bench.py r5g 7200 10800 11952 14976 18000 18700 29988 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tee $O/k4bench.txt
for rep in 1 2; do
for form in gemm library; do
  VLPET_AB=1 VLPET_K4_FORM=$form timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_bart_k4_${form}_$rep.json.log 2>&1
  python - $O/bench_bart_k4_${form}_$rep.json.log $form <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not l: print(sys.argv[2], "no json"); raise SystemExit
j=json.loads(l[-1]); k=j["kernels"]
print(sys.argv[2], j["value"], j["ms_per_step"], {n:(k[n]["avg_us"]) for n in ("k4_fwd","k4_ln_bwd","k4_wgrad") if n in k})
PY
done; done 2>&1 | tee $O/k4_instep_ab.txt
