#!/bin/bash
# round 2, pass ah: cycle stamps -- K1 backward rows kernel after the loader-chain change, K1 forward (down / up stage, chain A / chain G)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ah; mkdir -p $O
export VLPET_LIB=$GRAFT_REPO_ROOT/vl-pet_amd/lib/libvlpet_hip_stamps.so
{
echo "== bwd rows, chain A"; VLPET_DBG=16 python tools/kfwd_only.py 28000 bwd 2>&1 | grep "bwd2 ts" | tail -1
echo "== bwd rows, chain G"; VLPET_DBG=144 python tools/kfwd_only.py 28000 bwd 2>&1 | grep "bwd2 ts" | tail -1
echo "== fwd (training form), chain A, down stage 5"; VLPET_DBG=16 python tools/kfwd_only.py 28000 fwds 2>&1 | grep "vlpet ts" | tail -1
echo "== fwd, chain G, down stage 5"; VLPET_DBG=144 python tools/kfwd_only.py 28000 fwds 2>&1 | grep "vlpet ts" | tail -1
echo "== fwd, chain A, up stage 5"; VLPET_DBG=48 python tools/kfwd_only.py 28000 fwds 2>&1 | grep "vlpet ts" | tail -1
echo "== fwd, chain G, up stage 5"; VLPET_DBG=176 python tools/kfwd_only.py 28000 fwds 2>&1 | grep "vlpet ts" | tail -1
} | tee $O/stamps.txt
