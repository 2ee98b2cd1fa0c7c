#!/bin/bash
# round 6, GPU pass ck2: fused cross-attention keys in the T5 host: tests, ABBA in the T5 step (full batch and rank 1 of 8)
O=gpurun_out/r6ck2; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_crosskeys.py tests/test_host_golden.py tests/test_gpu_graph.py tests/test_gpu_modules.py -q -x 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest.txt
for tag in ck_a nock_a nock_b ck_b; do
  case $tag in nock*) export VLPET_NO_FUSED_CROSS_KEYS=1;; *) unset VLPET_NO_FUSED_CROSS_KEYS;; esac
  VLPET_AB=1 timeout 600 python bench.py --model t5 --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_t5_$tag.json.log 2>&1
  VLPET_AB=1 timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 40 --warmup 6 --no-cpu-baseline > $O/bench_t5r8_$tag.json.log 2>&1
done
unset VLPET_NO_FUSED_CROSS_KEYS
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6ck2/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l)
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], j.get("ab_switches"))
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
