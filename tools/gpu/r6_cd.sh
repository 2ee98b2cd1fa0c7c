#!/bin/bash
# round 6, GPU pass cd: the joint encoder's cat + dropout as one pass each way: tests (its own, the model-level suites), ABBA in the step
O=gpurun_out/r6cd; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_act.py tests/test_host_golden.py tests/test_gpu_graph.py tests/test_gpu_modules.py tests/test_gpu_dp.py tests/test_gpu_crosskeys.py tests/test_gpu_video.py -q -x 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest.txt
for tag in cd_a nocd_a nocd_b cd_b; do
  case $tag in nocd*) export VLPET_NO_CONCAT_DROPOUT=1;; *) unset VLPET_NO_CONCAT_DROPOUT;; esac
  VLPET_AB=1 timeout 600 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_$tag.json.log 2>&1
  VLPET_AB=1 timeout 600 python bench.py --model t5 --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_t5_$tag.json.log 2>&1
  VLPET_AB=1 timeout 600 python bench.py --emulate-ranks 8 --steps 40 --warmup 6 --no-cpu-baseline > $O/bench_r8_$tag.json.log 2>&1
done
unset VLPET_NO_CONCAT_DROPOUT
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6cd/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], {n: k[n]["avg_us"] for n in ("concat_drop_fwd", "concat_drop_bwd") if n in k}, j.get("ab_switches"))
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
