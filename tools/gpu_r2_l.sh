#!/bin/bash
# round 2, pass l: timing experiments on cols3 (wrong results on purpose): where does pass 2 wait?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
O=gpurun_out/r2l
for f in 3 4 5 6 7; do
  VLPET_BWD3=1 VLPET_BWD3_FORM=$f timeout 300 python tools/kbench.py 28000 bf16 > $O/kbench_28000_form$f.txt 2>&1
  echo "form $f"; grep -E "two-pass" $O/kbench_28000_form$f.txt
done
