#!/bin/bash
# round 6, GPU pass c: in-launch reduce-scatter in K1 (r <= 96) and K2 / K3 pass 2 -- parity suites, then an ABBA in-step A/B against the
# finalize-launch form (diagnosis build, VLPET_COLS_RED=1|0: same binary, same box)
O=gpurun_out/r6c; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_ng.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
DBG=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for arm in 1 0 0 1; do
  VLPET_LIB=$DBG VLPET_COLS_RED=$arm timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_red${arm}_$RANDOM.json.log 2>&1
done
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6c/bench_red*.json.log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], "steady", j["steady_state"]["value"], "op_us", j["roofline"].get("op_avg_us"), "frac", j["roofline"]["frac"],
                  {n: k[n]["avg_us"] for n in ("k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k2_bwd", "k2_fwd") if n in k})
P
