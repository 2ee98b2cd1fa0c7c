#!/bin/bash
# round 2, pass bs (fourth session, final): whole GPU suite, smoke, default bench + rocprofv3 kernel statistics (BART and T5, T5 with
# the norm hand-over off and on), the other workloads' lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bs; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 > $O/pytest_gpu.txt; tail -2 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_default.json.log 2> $O/bench.err
prof() {  # name, extra bench args...
  n=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$n -o $n -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 8 --warmup 4 --no-cpu-baseline --kernel-table off > $GRAFT_REPO_ROOT/$O/prof_$n.log 2>&1 )
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${n}_kernel_stats.csv
  rm -rf $O/prof_$n
}
prof bart
prof t5 --model t5
VLPET_NO_NORM_LINK=1 prof t5_nolink --model t5
timeout 600 python bench.py --model lora --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_lora.json.log 2>$O/l.err
timeout 600 python bench.py --model t5 --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_t5.json.log 2>$O/t.err
timeout 600 python bench.py --model video --steps 12 --warmup 4 --no-cpu-baseline --kernel-table off > $O/bench_video.json.log 2>$O/v.err
timeout 600 python bench.py --gpus 2 --backend gloo --scaling strong --steps 8 --warmup 3 --no-cpu-baseline --kernel-table off > $O/bench_dp2_gloo_strong.json.log 2>$O/d.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2bs/bench_*.json.log")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f.split('/')[-1], j["value"], j["unit"], j["ms_per_step"], "ms  n_gpus", j["n_gpus"], "roofline", r.get("kernel"), "frac", r.get("frac"), "op_frac", r.get("op_frac"), "op_us", r.get("op_avg_us"))
    except Exception as e: print(f, "ERR", e)
PY
