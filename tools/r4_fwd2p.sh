#!/bin/bash
# round 4, first GPU run: parity of the two-pass forward, then same-box A/B against the one-kernel forward (debug build switches)
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
export HIP_FORCE_DEV_KERNARG=1
echo "== sanity (small M, short timeout)" | tee $O/log.txt
timeout 180 python - <<'PY' 2>&1 | tee -a $O/log.txt
import sys, torch
sys.path.insert(0, "tests")
import gpu_cases as C
for (M, r) in [(32, 96), (224, 96), (1000, 96), (224, 192), (3000, 192), (224, 32), (5000, 96)]:
    e = C.run_k1(torch.bfloat16, M=M, d=768, r=r, rg=r, nh=4)
    print(M, r, {k: f"{v:.2e}" for k, v in e.items()}, flush=True)
    assert max(e.values()) <= 1e-2, e
print("sanity ok")
PY
echo "== K1 gpu tests" | tee -a $O/log.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_cols.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py tests/test_gpu_video.py -m gpu -x -q 2>&1 | tail -15 | tee -a $O/log.txt
echo "== forward A/B (debug build)" | tee -a $O/log.txt
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so K1BENCH_FWD_ONLY=1
for rnd in 1 2; do
for mode in 0 1 2 3; do
  VLPET_FWD2P=$mode K1BENCH_R=96 timeout 300 python tools/k1bench.py fwd2p=$mode 2100 3500 15272 28000 31616 33200 46648 2>&1 | grep k1bench | tee -a $O/k1fwd_r96.txt
done
for mode in 0 1 2 3; do
  VLPET_FWD2P=$mode K1BENCH_R=192 timeout 300 python tools/k1bench.py fwd2p=$mode 2100 3500 9200 16800 18250 28000 2>&1 | grep k1bench | tee -a $O/k1fwd_r192.txt
done
done
echo "== cold inputs" | tee -a $O/log.txt
unset K1BENCH_FWD_ONLY
