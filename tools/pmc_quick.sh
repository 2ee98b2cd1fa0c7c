#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <mode> [M]: two rocprofv3 --pmc passes (wave cycles / waits, LDS conflicts) + kernel stats of tools/kfwd_only.py
tag=$1; mode=${2:-bwd}; M=${3:-28000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag -o p$i -- python tools/kfwd_only.py $M $mode > gpurun_out/pmc_$tag/log$i.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_$tag -o kt -- python tools/kfwd_only.py $M $mode > /dev/null 2>&1
cut -d, -f1-4 gpurun_out/pmc_$tag/kt_kernel_stats.csv | grep -v "at::native" | head -6
