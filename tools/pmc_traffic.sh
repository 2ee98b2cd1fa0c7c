#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> <mode> [M]: FETCH_SIZE / WRITE_SIZE / L2 hit passes (separate rocprofv3 --pmc runs, kernel-trace only)
tag=$1; mode=${2:-bwd}; M=${3:-28000}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic_$tag
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/traffic_$tag -o p$i -- python tools/kfwd_only.py $M $mode > gpurun_out/traffic_$tag/log$i.txt 2>&1
done
python tools/pmc_summary.py gpurun_out/traffic_$tag ""
