#!/bin/bash
# round 6, GPU pass f: pass 2 of the K1 backward with the down side's dx1 / dx2 rows staged through LDS (64-byte runs of 16 rows per store
# instruction) vs lane-per-row 16-byte stores (libvlpet_hip_ab.so = the same sources with -DVLPET_COLS_STAGED=0): parity, then ABBA
O=gpurun_out/r6f; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_cols.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/pytest.txt
AB=$PWD/vl-pet_amd/lib/libvlpet_hip_ab.so
for arm in ab new new ab; do
  if [ $arm = ab ]; then export VLPET_LIB=$AB; else unset VLPET_LIB; fi
  python tools/k1red.py 15272 28000 31616 46648 2>&1 | grep -v amdgpu.ids | sed "s/^/$arm /" | tee -a $O/k1red.txt
done
unset VLPET_LIB
for arm in ab new new ab; do
  if [ $arm = ab ]; then export VLPET_LIB=$AB; else unset VLPET_LIB; fi
  timeout 600 python bench.py --steps 24 --warmup 4 --no-cpu-baseline > $O/bench_${arm}_$RANDOM.json.log 2>&1
done
unset VLPET_LIB
python - <<'P' | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r6f/bench_*.json.log")):
    ok = False
    for l in open(f):
        if l.startswith("{"):
            ok = True
            j = json.loads(l); k = j["kernels"]
            print(f.split("/")[-1], j["value"], "steady", j["steady_state"]["value"], "op_us", j["roofline"].get("op_avg_us"), "frac", j["roofline"]["frac"],
                  {n: k[n]["avg_us"] for n in ("k1_bwd_rows", "k1_bwd_wgrad", "k1_bwd_fin") if n in k}, {t: v["median"] for t, v in j["step_ms_by_task"].items()})
    if not ok: print(f, "NO JSON", open(f).read()[-800:])
P
