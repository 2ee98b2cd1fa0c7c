#!/bin/bash
# round 2, pass at: weight-gradient chunks capped at 1,408 rows: parity of everything that uses the plan, then in-step A/B (cap vs no cap)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2at; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_k4.py tests/test_gpu_lowrank.py tests/test_gpu_video.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest.txt
for i in 1 2; do
  VLPET_WGRAD_MAXROWS=0 timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_nocap_$i.json.log 2>$O/a$i.err
  timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_cap_$i.json.log 2>$O/b$i.err
done
VLPET_WGRAD_MAXROWS=0 timeout 600 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_video_nocap.json.log 2>$O/c.err
timeout 600 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_video_cap.json.log 2>$O/d.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2at/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j.get("kernels",{})
        g=lambda n: k.get(n,{}).get("avg_us")
        print(f.split('/')[-1], j["value"], j["ms_per_step"], "op", j["roofline"].get("op_avg_us"), "k1_wgrad", g("k1_bwd_wgrad"), "k2_bwd", g("k2_bwd"), "k4_wgrad", g("k4_wgrad"))
    except Exception as e: print(f, "ERR", e)
PY
