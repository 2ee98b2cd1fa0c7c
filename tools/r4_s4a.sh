#!/bin/bash
# after the attention rewrite (branch-free elementwise work, whole-line stores): whole GPU suite + the four bench lines
O=gpurun_out/r4bd; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
python bench.py --no-cpu-baseline > $O/bench_bart.json.log 2> $O/bench_bart.err
python bench.py --model t5 --no-cpu-baseline > $O/bench_t5.json.log 2> $O/bench_t5.err
python bench.py --model lora --no-cpu-baseline > $O/bench_lora.json.log 2> $O/bench_lora.err
python bench.py --emulate-ranks 8 --no-cpu-baseline > $O/bench_bart_rank1of8_graph.json.log 2> $O/bench_bart_rank1of8.err
for f in bart t5 lora bart_rank1of8_graph; do python - <<P
import json
d=json.loads(open("$O/bench_$f.json.log").read().strip().splitlines()[-1])
print("$f", d["value"], d["ms_per_step"], d["roofline"].get("frac"), {k:v for k,v in d.get("kernels_us",{}).items() if "attn" in k})
P
done
