#!/bin/bash
# round 2, pass ba (fourth session): dgrad GEMMs accumulate onto parked gradients (functional.linear_acc) -- parity of the
# new path (layer A/B tests, module + whole-model goldens, DP), then same-box bench A/B with the hand-over off / on
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ba; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_dp.py tests/test_gpu_video.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_subset.txt; tail -3 $O/pytest_subset.txt
VLPET_NO_GEMM_LINK=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_nolink.json.log 2> $O/b0.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_link.json.log 2> $O/b1.err
VLPET_NO_GEMM_LINK=1 timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_nolink2.json.log 2> $O/b2.err
timeout 600 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_link2.json.log 2> $O/b3.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2ba/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e); print(open(f.replace('.json.log','.err') if False else f).read()[-500:])
PY
tail -5 $O/b1.err
