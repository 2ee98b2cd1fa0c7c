#!/bin/bash
# round 2, pass av: weight-gradient plan bounded by the XCD geometry (no 33rd workgroup on a 32-CU XCD): parity, cold microbenchmark,
# in-step A/B against the previous plan (an explicit VLPET_WGRAD_WGS=256 bypasses the bound = the old behaviour)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2av; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_k4.py tests/test_gpu_lowrank.py tests/test_gpu_video.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/pytest.txt
{
for M in 24000 31616 46648; do
  echo "-- new plan";  K1BENCH_COLD=1 timeout 200 python tools/k1bench.py new $M | sed -E 's/fwd\+save +[0-9.]+ us +bwd rows +[0-9.]+ us +//'
  echo "-- old plan";  K1BENCH_COLD=1 VLPET_WGRAD_WGS=256 timeout 200 python tools/k1bench.py old $M | sed -E 's/fwd\+save +[0-9.]+ us +bwd rows +[0-9.]+ us +//'
  echo "-- 512 target"; K1BENCH_COLD=1 VLPET_WGRAD_WGS=512 timeout 200 python tools/k1bench.py w512 $M | sed -E 's/fwd\+save +[0-9.]+ us +bwd rows +[0-9.]+ us +//'
done
} 2>&1 | grep -v amdgpu.ids | tee $O/k1bench_cold_plan.txt
for i in 1 2; do
  VLPET_WGRAD_WGS=256 timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_old_$i.json.log 2>$O/a$i.err
  timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_new_$i.json.log 2>$O/b$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2av/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j.get("kernels",{})
        g=lambda n: k.get(n,{}).get("avg_us")
        print(f.split('/')[-1], j["value"], j["ms_per_step"], "op", j["roofline"].get("op_avg_us"), "op_frac", j["roofline"].get("op_frac"), "k1_wgrad", g("k1_bwd_wgrad"), "k2_bwd", g("k2_bwd"), "k4_wgrad", g("k4_wgrad"))
    except Exception as e: print(f, "ERR", e)
PY
