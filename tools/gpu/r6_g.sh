#!/bin/bash
# round 6, GPU pass g (evidence, first set): PMC traffic of the K1 backward as it runs now (in-launch reduce-scatter from 8,192 rows) at the four
# task sizes + r = 192 + K5 / K1 forward / K4; rocprofv3 kernel statistics of the BART / T5 bench commands; bench lines of the other configs
O=gpurun_out/r6g; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
bash tools/pmc_traffic4.sh $O/pmc "k1bwd 28000" "k1bwd 46648" "k1bwd 15272" "k1bwd 31616" "k1bwd 3500" "k1bwd 18250 192" "k1bwd 28000 192" "k5fwd 28000" "k5bwd 28000" "k1fwd 28000" "k4fwd 18700"
cp $O/pmc/summary.txt $O/pmc_summary.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bart -o kt -- python bench.py --steps 20 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_bart_prof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t5 -o kt -- python bench.py --model t5 --steps 12 --warmup 4 --kernel-table off --no-cpu-baseline > $O/bench_t5_prof.log 2>&1
for m in bart t5; do f=$(find $O/prof_$m -name "kt_kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/kernel_stats_$m.csv; rm -rf $O/prof_$m; done
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_lora.json.log 2>&1
timeout 600 python bench.py --model video --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_video.json.log 2>&1
timeout 600 python bench.py --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_bart_rank1of8.json.log 2>&1
timeout 600 python bench.py --model t5 --emulate-ranks 8 --steps 20 --warmup 6 --no-cpu-baseline > $O/bench_t5_rank1of8.json.log 2>&1
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6g/bench_*.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], j["ms_per_step"], "median", j["step_ms_median"], "steady", j["steady_state"]["value"], "peak GB", j.get("peak_memory_GB"), "frac", j["roofline"]["frac"], "op_us", j["roofline"].get("op_avg_us"),
                  {n: k[n]["avg_us"] for n in ("k1_fwd", "k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin", "k5_fwd", "k5_bwd", "k4_fwd", "k2_bwd", "k3_fwd", "k3_bwd") if n in k})
    if not ok: print(f, "NO JSON"); print(open(f).read()[-1200:])
P
ls -la $O
