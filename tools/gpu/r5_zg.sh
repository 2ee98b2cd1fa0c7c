#!/bin/bash
# round 5, GPU pass zg: K1's finalize launch as a parallel branch of the captured graph (functional.FINALIZE_SIDE_STREAM) -- graph tests, A/B in the replayed steps
O=gpurun_out/r5zg; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_host_golden.py -m gpu -q 2>&1 | tail -6 | tee $O/pytest.txt
for rep in 1 2; do
  for m in bart t5; do
    extra=""; [ $m = t5 ] && extra="--model t5"
    timeout 600 python bench.py $extra --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_${m}_rank1of8_side_$rep.json.log 2>&1
    VLPET_AB=1 VLPET_NO_FINALIZE_SIDE_STREAM=1 timeout 600 python bench.py $extra --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_${m}_rank1of8_inline_$rep.json.log 2>&1
  done
  timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_side_$rep.json.log 2>&1
  VLPET_AB=1 VLPET_NO_FINALIZE_SIDE_STREAM=1 timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_t5_inline_$rep.json.log 2>&1
done
python - <<'P' | tee gpurun_out/r5zg/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r5zg/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); ok = True
            print(f.split("/")[-1], j["value"], j["ms_per_step"], j.get("step_mode", "")[:40])
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1500:])
P
