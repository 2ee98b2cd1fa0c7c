#!/bin/bash
# round 2, pass x: where the time of the streaming weight-gradient kernel and of the finalize goes (experiment build, one process)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x; mkdir -p $O
VLPET_LIB=$GRAFT_REPO_ROOT/vl-pet_amd/lib/libvlpet_hip_exp.so python tools/wgbench.py 28000 modes > $O/modes2.txt 2>&1
cat $O/modes2.txt | grep -v amdgpu.ids
# ground truth: the bench with the per-kernel brackets, new kernel vs the previous one, same box
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_stream.json.log 2>$O/bench_stream.err
VLPET_WGRAD_STREAM=0 timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_old.json.log 2>$O/bench_old.err
python - <<'PY'
import json
for f in ("bench_stream","bench_old"):
    try:
        j=json.loads(open(f"gpurun_out/r2x/{f}.json.log").read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["roofline"]); print({n:v["avg_us"] for n,v in j.get("kernels",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o bart -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/prof_bart.log 2>&1
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O/prof_bart -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -i "wgrad\|pet_\|tail\|visproj" $f | cut -c1-140
