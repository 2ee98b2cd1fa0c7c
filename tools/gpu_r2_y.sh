#!/bin/bash
# round 2, pass y: what bounds the streaming weight-gradient kernel (probe modes, ring depth, workgroups per CU, counters)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
E=$GRAFT_REPO_ROOT/vl-pet_amd/lib/libvlpet_hip_exp.so
VLPET_LIB=$E python tools/wgbench.py 28000 modes 2>&1 | grep -v amdgpu.ids > $O/modes_default.txt
VLPET_LIB=$E VLPET_WGRAD_WGS=512 python tools/wgbench.py 28000 modes 2>&1 | grep -v amdgpu.ids > $O/modes_wgs512.txt
VLPET_LIB=$E VLPET_WGRAD_NSTG=4 python tools/wgbench.py 28000 modes 2>&1 | grep -v amdgpu.ids > $O/modes_nstg4.txt
VLPET_LIB=$E VLPET_WGRAD_WGS=128 python tools/wgbench.py 28000 modes 2>&1 | grep -v amdgpu.ids > $O/modes_wgs128.txt
for f in default wgs512 nstg4 wgs128; do echo "== $f"; grep "reference\|rep 1" $O/modes_$f.txt; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc -o p$i -- python tools/wgbench.py 28000 pmc > $O/pmc_log$i.txt 2>&1
done
python tools/pmc_summary.py $O/pmc wgrad 2>&1 | cut -c1-200 | tee $O/pmc_summary.txt
find $O -name "*_kernel_trace.csv" -size +2M -delete
