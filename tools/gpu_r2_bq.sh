#!/bin/bash
# round 2, pass bq: bias gradients of the LoRA runs through vlpet_colsum -- parity, then same-box LoRA bench A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bq; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_host_golden.py tests/test_gpu_modules.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/pytest.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "lora or k3" 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/pytest.txt
for i in 1 2; do
VLPET_NO_BIAS_GRAD_KERNEL=1 timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_autograd_$i.json.log 2>$O/l0.err
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_colsum_$i.json.log 2>$O/l1.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bq/bench_*.json.log")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j["value"], j["ms_per_step"])
PY
