#!/bin/bash
# round 2, pass aj: weight gradients on a side stream (Trainer(overlap_wgrad=True)) with the streaming kernel: same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aj; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_default.json.log 2>$O/a.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --overlap-wgrad > $O/bench_overlap.json.log 2>$O/b.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --overlap-wgrad --kernel-table off > $O/bench_overlap_notable.json.log 2>$O/c.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_default2.json.log 2>$O/d.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2aj/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"].get("op_frac"), {n:v["avg_us"] for n,v in j.get("kernels",{}).items() if n.startswith("k1_bwd")})
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $O/b.err
