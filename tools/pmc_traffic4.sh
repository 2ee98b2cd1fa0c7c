#!/bin/bash
# usage: tools/pmc_traffic4.sh <out_dir> "<op> <M> [r]" ...: FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 --pmc runs, kernel-trace only) + kernel stats
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $out
for spec in "$@"; do
  tag=$(echo $spec | tr ' ' '_')
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -o p$i -- python tools/pmc_target.py $spec > $out/$tag.log$i.txt 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -o kt -- python tools/pmc_target.py $spec > /dev/null 2>&1
  echo "== $spec" >> $out/summary.txt
  python tools/pmc_summary.py $out/$tag "" 2>/dev/null | grep -v -E "at::native|elementwise|pack_pair|distribution" >> $out/summary.txt
  python - "$out/$tag" >> $out/summary.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/kt_kernel_stats.csv", recursive=True) + glob.glob(sys.argv[1] + "/kt_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("at::native", "elementwise", "pack_pair", "distribution")): continue
        print(f"   duration  {n[:70]:70s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:8.2f} us")
    break
PY
  rm -rf $out/$tag
done
