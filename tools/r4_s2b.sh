#!/bin/bash
# round 4, second session, run 2: finalize with non-temporal loads (same-box A/B of two builds), per-kernel times of the two-pass K2 / K3 forward
mkdir -p gpurun_out/r4n
O=gpurun_out/r4n
export HIP_FORCE_DEV_KERNARG=1
for rep in 1 2; do
for lib in libvlpet_hip_fin0.so libvlpet_hip.so; do
  VLPET_LIB=$PWD/vl-pet_amd/lib/$lib K1BENCH_R=96 timeout 300 python tools/k1bench.py $lib 3500 15272 28000 46648 2>&1 | grep k1bench | tee -a $O/k1bench_fin.txt
  VLPET_LIB=$PWD/vl-pet_amd/lib/$lib K2BENCH_R=96 timeout 300 python tools/k2bench.py $lib 3500 28000 2>&1 | grep k2bench | tee -a $O/k2bench_fin.txt
done; done
export VLPET_LIB=$PWD/vl-pet_amd/lib/libvlpet_hip_dbg.so
for mode in 0 1; do
  VLPET_FWD2P=$mode K2BENCH_R=96 timeout 300 python tools/k2bench.py fwd2p=$mode 20000 24000 28000 36000 40000 46648 2>&1 | grep k2bench | tee -a $O/k2bench.txt
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for M in 10000 28000; do
VLPET_FWD2P=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k3_$M -o kt -- python tools/k3bench.py $M > $O/prof_k3_$M.log 2>&1
f=$(find $O/prof_k3_$M -name "kt_kernel_stats.csv" | head -1)
echo "== M=$M" | tee -a $O/k3_kernels.txt
cut -d, -f1-4 $f | grep -E "drop_bits|k1_down|k1_up|pet_fwd_kernel" | tee -a $O/k3_kernels.txt
done
