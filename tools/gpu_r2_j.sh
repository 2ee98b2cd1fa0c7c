#!/bin/bash
# round 2, pass j: cols2 (two waves per tile, builtin MFMAs) parity + timing + SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j
for f in 1 2; do
  VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -k "k1 or gate or encoder or residual" > $O/pytest_form$f.log 2>&1; echo "rc=$?" >> $O/pytest_form$f.log
  tail -4 $O/pytest_form$f.log
done
for f in 1 2; do
  VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 300 python tools/kbench.py 28000 bf16 > $O/kbench_28000_form$f.txt 2>&1
  grep -E "two-pass|previous" $O/kbench_28000_form$f.txt
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $O/pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_WAVE32_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  VLPET_BWD3=1 VLPET_BWD3_FORM=2 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc -o bwd$i -- python tools/kfwd_only.py 28000 bwd > $O/pmc/log_bwd$i.txt 2>&1
done
python tools/pmc_summary.py $O/pmc "" > $O/pmc_summary_cols2.txt 2>&1
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
cat $O/pmc_summary_cols2.txt | grep -E "cols2|dz_kernel" | head -60
