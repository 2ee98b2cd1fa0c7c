#!/bin/bash
# round 5, GPU pass a: K5 rewrite (8-byte pieces, LDS parameters, nt loads) -- parity + A/B against the round-4 library
O=gpurun_out/r5a; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_k4.py tests/test_gpu_lowrank.py tests/test_gpu_video.py -m gpu -q -s 2>&1 | grep -E "max\|beta|passed|failed|Error|error" | tail -30 | tee $O/pytest.txt
SZ="2500 3500 10000 16640 28000 30384 46648"
for rep in 1 2; do
  echo "== r4 lib" | tee -a $O/k5abi.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_r4.so python tools/k5abi.py $SZ 2>&1 | tee -a $O/k5abi.txt
  echo "== r5 lib" | tee -a $O/k5abi.txt
  python tools/k5abi.py $SZ 2>&1 | tee -a $O/k5abi.txt
done
for cap in 512 768 1024 1280; do
  echo "== r5 dbg cap $cap" | tee -a $O/k5abi.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so VLPET_DBG=$cap python tools/k5abi.py 10000 28000 46648 2>&1 | tee -a $O/k5abi.txt
done
timeout 600 python bench.py --steps 10 --warmup 4 > $O/bench_bart.json.log 2>&1; tail -c 3000 $O/bench_bart.json.log
