#!/bin/bash
# round 5, GPU pass b: K4 forward as a tiled GEMM with exchanged LayerNorm statistics -- parity, timing of the ring forms, graph-mode tests
O=gpurun_out/r5b; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 900 python -m pytest tests/test_gpu_k4.py -m gpu -q -x 2>&1 | tail -15 | tee $O/pytest_k4.txt
timeout 300 python tools/k4bench.py r5 7200 10800 14976 18000 18700 29988 2>&1 | grep -v amdgpu.ids | tee $O/k4bench.txt
timeout 900 python -m pytest tests/test_gpu_graph.py tests/test_gpu_modules.py tests/test_host_golden.py tests/test_gpu_video.py -m gpu -q 2>&1 | tail -8 | tee $O/pytest_graph.txt
timeout 600 python bench.py --steps 10 --warmup 4 > $O/bench_bart.json.log 2>&1; tail -c 1500 $O/bench_bart.json.log
