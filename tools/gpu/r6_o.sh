#!/bin/bash
# round 6, GPU pass o: the position / order branch kernel (csrc/vispos.hip): its tests, the K4 suite, an ABBA in-step A/B against the
# library-op chain, and the ABBA in-step A/B of the K4 GEMM forms the round-5 verdict asked for (gemm, library, library, gemm)
O=gpurun_out/r6o; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py tests/test_gpu_lowrank.py tests/test_host_golden.py -q -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/pytest_k4.txt
for tag in pos_a nopos_a nopos_b pos_b; do
  case $tag in nopos*) export VLPET_NO_POS_KERNEL=1;; *) unset VLPET_NO_POS_KERNEL;; esac
  VLPET_AB=1 timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_$tag.json.log 2>&1
done
unset VLPET_NO_POS_KERNEL
for tag in gemm_a library_a library_b gemm_b; do
  VLPET_AB=1 VLPET_K4_FORM=${tag%_*} timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_k4_$tag.json.log 2>&1
done
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6o/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "eager_region", j.get("eager_region", {}).get("ms_per_step"),
                  {n: k[n]["avg_us"] for n in ("k4_fwd", "k4_ln_bwd", "k4_wgrad", "k4_pos_fwd", "k4_pos_bwd") if n in k}, j.get("ab_switches"))
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
