#!/bin/bash
# round 2, pass ad: cycle stamps of the chain-split K1 backward rows kernel (one wave of workgroup 0), chain A and chain G, M = 28000 and 128
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ad; mkdir -p $O
export VLPET_LIB=$GRAFT_REPO_ROOT/vl-pet_amd/lib/libvlpet_hip_stamps.so
for M in 28000 128; do
  echo "== M=$M chain A"; VLPET_DBG=16 python tools/kfwd_only.py $M bwd 2>&1 | grep "bwd2 ts" | tail -2
  echo "== M=$M chain G"; VLPET_DBG=144 python tools/kfwd_only.py $M bwd 2>&1 | grep "bwd2 ts" | tail -2
done | tee $O/stamps.txt
