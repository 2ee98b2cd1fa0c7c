#!/bin/bash
# round 2, pass z3: finalize with host-built arguments: parity, pair timeline, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_k4.py tests/test_gpu_video.py tests/test_gpu_gates.py -m gpu -q -x > $O/pytest_sub.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sub.log
tail -3 $O/pytest_sub.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for dk in 0 1; do
  HIP_FORCE_DEV_KERNARG=$dk timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trb$dk -o t -- python tools/wgbench.py 28000 x > $O/trace_b_dk$dk.txt 2>&1
  echo "== HIP_FORCE_DEV_KERNARG=$dk"; grep "wgbench" $O/trace_b_dk$dk.txt
  f=$(find $O/trb$dk -name "*kernel_trace.csv" | head -1)
  python tools/trace_pairs.py $f 100 0 | head -3 | tee $O/pairs_b_dk$dk.txt
  rm -f $f
done
for dk in 0 1; do
  HIP_FORCE_DEV_KERNARG=$dk timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_b_dk$dk.json.log 2>$O/bench_b_dk$dk.err
done
python - <<'PY'
import json
for f in ("bench_b_dk0","bench_b_dk1"):
    try:
        j=json.loads(open(f"gpurun_out/r2z/{f}.json.log").read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"]); print({n:v["avg_us"] for n,v in j.get("kernels",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
