#!/bin/bash
# round 5, GPU pass n: K1 backward from the forward's output (vlpet_adapter_gate_bwd_saved_y) -- parity, A/B through the C ABI and in
# the step; K5 workgroup count at the per-rank sizes (debug build: VLPET_DBG = cap of tail_blocks)
O=gpurun_out/r5n; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests/test_gpu_cols.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest.txt
for rep in 1 2; do
  for r in 96 192; do
    SZ="3500 8232 17000 28000 33200"; [ $r = 192 ] && SZ="2128 8000 18250 28000"
    K1BENCH_R=$r K1BENCH_FROM_X2=1 python tools/k1bench.py from_x2 $SZ 2>&1 | tee -a $O/k1bench.txt
    K1BENCH_R=$r python tools/k1bench.py from_y $SZ 2>&1 | tee -a $O/k1bench.txt
  done
done
K1BENCH_COLD=1 K1BENCH_FROM_X2=1 python tools/k1bench.py from_x2 28000 2>&1 | tee -a $O/k1bench.txt
K1BENCH_COLD=1 python tools/k1bench.py from_y 28000 2>&1 | tee -a $O/k1bench.txt
K1BENCH_R=192 K1BENCH_COLD=1 K1BENCH_FROM_X2=1 python tools/k1bench.py from_x2 18250 2>&1 | tee -a $O/k1bench.txt
K1BENCH_R=192 K1BENCH_COLD=1 python tools/k1bench.py from_y 18250 2>&1 | tee -a $O/k1bench.txt
for cap in 128 256 512 1024; do
  echo "== dbg cap $cap" | tee -a $O/k5abi_small.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_DBG=$cap python tools/k5abi.py 500 1000 2000 3500 5000 6000 8000 2>&1 | cut -c1-420 | tee -a $O/k5abi_small.txt
done
for rep in 1 2; do
  timeout 600 python bench.py --steps 16 --warmup 5 --no-cpu-baseline > $O/bench_bart_y_$rep.json.log 2>&1
  VLPET_AB=1 VLPET_K1_BWD_FROM_X2=1 timeout 600 python bench.py --steps 16 --warmup 5 --no-cpu-baseline > $O/bench_bart_x2_$rep.json.log 2>&1
  timeout 600 python bench.py --model t5 --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_t5_y_$rep.json.log 2>&1
  VLPET_AB=1 VLPET_K1_BWD_FROM_X2=1 timeout 600 python bench.py --model t5 --steps 10 --warmup 4 --no-cpu-baseline > $O/bench_t5_x2_$rep.json.log 2>&1
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5n/bench_*.json.log")):
    for l in open(f):
        if l.startswith("{"):
            j = json.loads(l); k = j["kernels"]
            print(f, j["value"], j["ms_per_step"], "k1_bwd_rows", k["k1_bwd_rows"]["avg_us"], "op", j["roofline"]["op_avg_us"], j["roofline"]["frac"])
P
