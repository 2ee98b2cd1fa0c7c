#!/bin/bash
# round 2, pass al: dropout-mask generator: Philox-4x32-7 per 8 elements vs two multiply-xorshift rounds per element pair (same-box A/B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2al; mkdir -p $O
L=$GRAFT_REPO_ROOT/vl-pet_amd/lib
{
echo "== philox"; python tools/k5bench.py 28000; python tools/k3bench.py 28000 bf16 2>&1 | grep -i "r=64\|r = 64" | head -6
echo "== mix32"; VLPET_LIB=$L/libvlpet_hip_mix.so python tools/k5bench.py 28000; VLPET_LIB=$L/libvlpet_hip_mix.so python tools/k3bench.py 28000 bf16 2>&1 | grep -i "r=64\|r = 64" | head -6
} 2>&1 | grep -v amdgpu.ids | tee $O/rng_ab.txt
VLPET_LIB=$L/libvlpet_hip_mix.so timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_mix.json.log 2>$O/a.err
timeout 600 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_philox.json.log 2>$O/b.err
VLPET_LIB=$L/libvlpet_hip_mix.so timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_mix.json.log 2>$O/c.err
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora_philox.json.log 2>$O/d.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2al/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
