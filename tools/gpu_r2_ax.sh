#!/bin/bash
# round 2, pass ax: non-temporal policy on the once-read row streams (global_load_lds aux = 2) and on the rows kernel's final dx stores:
# three builds side by side (VLPET_LIB), cold microbenchmark, then in the step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ax; mkdir -p $O
L=$GRAFT_REPO_ROOT/vl-pet_amd/lib
{
for M in 28000 46648; do
  for v in hip hip_nt hip_ntdx; do echo "-- lib$v"; K1BENCH_COLD=1 VLPET_LIB=$L/libvlpet_$v.so timeout 200 python tools/k1bench.py $v $M; done
done
for v in hip hip_nt; do echo "-- warm lib$v"; VLPET_LIB=$L/libvlpet_$v.so timeout 200 python tools/k1bench.py $v 28000; done
} 2>&1 | grep -v amdgpu.ids | tee $O/k1bench_nt.txt
for i in 1 2; do
  for v in hip hip_nt hip_ntdx; do
    VLPET_LIB=$L/libvlpet_$v.so timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > $O/bench_${v}_$i.json.log 2>$O/$v$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ax/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); k=j.get("kernels",{})
        g=lambda n: k.get(n,{}).get("avg_us")
        print(f.split('/')[-1], j["value"], j["ms_per_step"], "op", j["roofline"].get("op_avg_us"), "rows", j["roofline"].get("avg_launch_us"), "k1_wgrad", g("k1_bwd_wgrad"), "k1_fwd", g("k1_fwd"), "k2_fwd", g("k2_fwd"), "k2_bwd", g("k2_bwd"))
    except Exception as e: print(f, "ERR", e)
PY
