#!/bin/bash
# round 2, pass ap: K1 weight gradients on a side stream (memory-bound kernel next to the backbone's compute-bound GEMMs), re-measured
# with the streaming weight-gradient kernel; same box, alternating A/B/A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ap; mkdir -p $O
for i in 1 2; do
  timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --kernel-table off > $O/bench_main_$i.json.log 2>$O/m$i.err
  timeout 600 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --kernel-table off --overlap-wgrad > $O/bench_side_$i.json.log 2>$O/s$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ap/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
