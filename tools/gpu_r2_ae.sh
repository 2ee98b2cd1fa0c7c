#!/bin/bash
# round 2, pass ae: static wave priorities in the two-chain K1 kernels (s_setprio for the gate-chain / loader waves): same-box A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ae; mkdir -p $O
L=$GRAFT_REPO_ROOT/vl-pet_amd/lib
{
python tools/k1bench.py default 28000 46648 15272 3500
for v in p1 p2 p3 p4; do VLPET_LIB=$L/libvlpet_hip_$v.so python tools/k1bench.py $v 28000 46648 15272 3500; done
python tools/k1bench.py default 28000
} 2>&1 | grep k1bench | tee $O/prio.txt
