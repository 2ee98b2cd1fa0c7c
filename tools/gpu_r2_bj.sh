#!/bin/bash
# round 2, pass bj: what bounds the short-sequence attention backward -- the product binary against a build whose backward only
# stages its images and stores (no MFMA / softmax work): tools/_lib_exp1.so (not committed; wrong results on purpose)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bj; mkdir -p $O
echo "== product" | tee $O/attn_exp.txt; timeout 300 python tools/attnbench2.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_exp.txt
echo "== backward without compute" | tee -a $O/attn_exp.txt; VLPET_LIB=$GRAFT_REPO_ROOT/tools/_lib_exp1.so timeout 300 python tools/attnbench2.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_exp.txt
