#!/bin/bash
# round 5, GPU pass f: in-step A/B of the K4 forms (same box, alternating), per-rank (emulated) lines with the exchange estimate
O=gpurun_out/r5f; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for form in gemm library; do
  VLPET_AB=1 VLPET_K4_FORM=$form timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_bart_k4_${form}_$rep.json.log 2>&1
  python - $O/bench_bart_k4_${form}_$rep.json.log $form <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")]
if not l: print(sys.argv[2], "no json"); raise SystemExit
j=json.loads(l[-1]); k=j["kernels"]
print(sys.argv[2], j["value"], j["ms_per_step"], {n:k[n]["avg_us"] for n in ("k4_fwd","k4_ln_bwd","k4_wgrad") if n in k})
PY
done; done 2>&1 | tee $O/k4_instep_ab.txt
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1; tail -c 2500 $O/bench_bart_rank1of8.json.log | head -c 1800; echo
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
timeout 900 python -m pytest tests/test_gpu_k4.py tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_cols.py -m gpu -q 2>&1 | tail -3 | tee $O/pytest.txt
