#!/bin/bash
# round 2, pass bu (projection nodes of the T5 attention) / br: T5 host, residual-stream gradient added inside the RMS norm's backward kernel -- parity, then same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bu; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_tail.py tests/test_host_golden.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest.txt
for i in 1 2; do
VLPET_NO_NORM_LINK=1 timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_t5_nolink_$i.json.log 2>$O/t0.err
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_t5_link_$i.json.log 2>$O/t1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bu/bench_*.json.log")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
PY
