#!/bin/bash
# tools/r05_profiles.sh [stats|pmc|all]: the rocprofv3 evidence for profiles/r05_* (run on the GPU box through gpurun).
# kernel-trace + stats per workload (bench.py's classes, the FEC and TX side benches), then the counter passes
# (tools/pmc.sh: separate --pmc runs, kernel trace only).
set -o pipefail
export TMPDIR=/tmp
WHAT=${1:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05prof
mkdir -p $O
cd /tmp
# config4 = the Transceiver group on the fused front end (bench.py's default); config4_stateless = the round-2 fused
# front end (TSC on every slot); config4_wideband = the channeliser in front of the group
wl_args() { case $1 in
  config4_stateless) echo "--workload config4 --stateless-frontend";;
  config4_unfused) echo "--workload config4 --unfused-frontend";;
  config4_wideband) echo "--workload config4 --wideband 8";;
  config4_reference_chain) echo "--workload config4 --reference-chain";;
  normal_exact) echo "--workload normal --soft-mode exact";;
  *) echo "--workload $1";; esac; }
if [ $WHAT != pmc ]; then
for W in normal normal_exact rach config4 config4_unfused config4_stateless config4_wideband config4_reference_chain config5; do
  echo "== stats $W"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py $(wl_args $W) --steps 200 --no-cpu-baseline --no-fresh --no-lever > $O/bench_under_rocprof_$W.json 2> $O/stats_$W.err || echo "stats $W failed"
  python3 $R/tools/prof_summary.py $O/stats_$W > $O/kernel_stats_$W.csv
  rm -rf $O/stats_$W                                       # (raw traces: gpurun brings back at most 64 MiB)
  cat $O/kernel_stats_$W.csv
done
for T in fec_bench txbe_bench group_tx_bench; do
  echo "== stats $T"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$T -- python3 $R/tools/$T.py > $O/$T.json 2> $O/stats_$T.err || echo "stats $T failed"
  python3 $R/tools/prof_summary.py $O/stats_$T > $O/kernel_stats_$T.csv
  rm -rf $O/stats_$T
  cat $O/kernel_stats_$T.csv
done
fi
cd $R
if [ $WHAT != stats ]; then
for W in normal normal_exact rach config4 config4_unfused config4_reference_chain config5; do
  echo "== pmc $W"
  bash tools/pmc.sh r05prof/pmc_$W $(wl_args $W) --no-fresh --no-lever > $O/pmc_$W.txt 2>&1 || echo "pmc $W failed"
  tail -3 $O/pmc_$W.txt
done
for T in fec_bench txbe_bench; do
  echo "== pmc $T"
  PMC_PROG=tools/$T.py bash tools/pmc.sh r05prof/pmc_$T > $O/pmc_$T.txt 2>&1 || echo "pmc $T failed"
  tail -3 $O/pmc_$T.txt
done
fi
