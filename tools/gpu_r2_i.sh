#!/bin/bash
# round 2, pass i: the two-waves-per-tile pass-2 kernel (pet_gate_cols2_kernel): parity under VLPET_BWD3=1, A/B timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
O=gpurun_out/r2i
for f in 1 2; do
  VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -x -k "k1 or gate or encoder or residual" > $O/pytest_form$f.log 2>&1; echo "rc=$?" >> $O/pytest_form$f.log
  tail -4 $O/pytest_form$f.log
done
for f in 0 1 2; do
  VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 300 python tools/kbench.py 28000 bf16 > $O/kbench_28000_form$f.txt 2>&1
  grep -E "two-pass|previous" $O/kbench_28000_form$f.txt
done
VLPET_BWD3=1 VLPET_BWD3_FORM=1 timeout 300 python tools/kbench.py 28000 fp32 > $O/kbench_28000_fp32_form1.txt 2>&1; grep -E "two-pass|previous" $O/kbench_28000_fp32_form1.txt
VLPET_BWD3=1 VLPET_BWD3_FORM=0 timeout 300 python tools/kbench.py 28000 fp32 > $O/kbench_28000_fp32_form0.txt 2>&1; grep -E "two-pass|previous" $O/kbench_28000_fp32_form0.txt
for M in 3500 512; do
 for f in 0 1; do VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 300 python tools/kbench.py $M bf16 > $O/kbench_${M}_form$f.txt 2>&1; grep -E "two-pass|previous" $O/kbench_${M}_form$f.txt; done
done
