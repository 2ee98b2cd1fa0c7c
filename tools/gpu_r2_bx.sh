#!/bin/bash
# round 2, pass bx: LoRA projections -- K3's input gradient taken over by the base projection's dgrad GEMM: parity, same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bx; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "lora or k3 or host_golden or tiny" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/pytest.txt
for i in 1 2; do
VLPET_NO_LORA_LINK=1 timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_nolink_$i.json.log 2>$O/l0.err
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_link_$i.json.log 2>$O/l1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bx/bench_*.json.log")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
PY
