#!/bin/bash
# round 6, GPU pass d: in-launch reduce-scatter (K1 r <= 96, K2 / K3) -- parity suites incl. held CUs, then ABBA in-step A/Bs against the
# finalize-launch form (product library, VLPET_AB=1 VLPET_FINALIZE_LAUNCH=0|1): full batch (eager) and emulated rank 1 of 8 (graph replay)
O=gpurun_out/r6d; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_k4.py tests/test_gpu_ng.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py tests/test_gpu_dp.py -x -q -s 2>&1 | grep -v amdgpu.ids | grep -E "held|passed|failed|Error|error|assert" | tail -20 | tee $O/pytest.txt
timeout 200 python tools/held_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/held_probe.txt
for arm in 0 1 1 0; do
  VLPET_AB=1 VLPET_FINALIZE_LAUNCH=$arm timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_full_fin${arm}_$RANDOM.json.log 2>&1
done
for arm in 0 1 1 0; do
  VLPET_AB=1 VLPET_FINALIZE_LAUNCH=$arm timeout 600 python bench.py --emulate-ranks 8 --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_rank8_fin${arm}_$RANDOM.json.log 2>&1
done
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6d/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], "median ms", j["step_ms_median"], "steady", j["steady_state"]["value"], "op_us", j["roofline"].get("op_avg_us"), "frac", j["roofline"]["frac"],
                  {n: k[n]["avg_us"] for n in ("k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k2_bwd") if n in k}, {t: v["median"] for t, v in j["step_ms_by_task"].items()})
    if not ok: print(f, "NO JSON", open(f).read()[-800:])
P
