#!/bin/bash
# round 6, GPU pass x3 (after the fused cross-attention keys and concat_dropout): the whole GPU suite + smoke + the default bench line at HEAD (after the diagnosis build was rebuilt)
O=gpurun_out/r6x3; mkdir -p $O
export HIP_FORCE_DEV_KERNARG=1
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 600 python bench.py > $O/bench_bart.json.log 2>&1
grep -o '"value": [0-9.]*, "unit": "samples/s", "n_gpus": 1' $O/bench_bart.json.log
