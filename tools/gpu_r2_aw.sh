#!/bin/bash
# round 2, pass aw: K5 kernels through the ABI, warm and cold, decoder- and encoder-sized launches; block-count sweep under cold inputs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aw; mkdir -p $O
{
timeout 300 python tools/k5abi.py 8320 10000 15272 28000 31616 46648
for B in 256 512 1024 1536 2048; do echo "== VLPET_TAIL_BLOCKS=$B"; VLPET_TAIL_BLOCKS=$B timeout 300 python tools/k5abi.py 10000 28000 46648; done
} 2>&1 | grep -v amdgpu.ids | tee $O/k5abi.txt
