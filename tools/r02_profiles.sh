#!/bin/bash
# tools/r02_profiles.sh: the rocprofv3 evidence for profiles/r02_* (run on the GPU box through gpurun).
# kernel-trace + stats per workload, then the counter passes (tools/pmc.sh: separate --pmc runs, kernel trace only).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02prof
mkdir -p $O
cd /tmp
# config4 = the fused front end (bench.py's default); config4_unfused = push + pop + detect through the resampled stream
wl_args() { case $1 in config4_unfused) echo "--workload config4 --unfused-frontend";; *) echo "--workload $1";; esac; }
for W in normal rach config4 config4_unfused config5; do
  echo "== stats $W"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py $(wl_args $W) --steps 200 --no-cpu-baseline --no-fresh > $O/bench_under_rocprof_$W.json 2> $O/stats_$W.err || echo "stats $W failed"
  python3 $R/tools/prof_summary.py $O/stats_$W > $O/kernel_stats_$W.csv
  cat $O/kernel_stats_$W.csv
done
cd $R
for W in normal rach config4 config4_unfused config5; do
  echo "== pmc $W"
  bash tools/pmc.sh r02prof/pmc_$W $(wl_args $W) --no-fresh > $O/pmc_$W.txt 2>&1 || echo "pmc $W failed"
  tail -3 $O/pmc_$W.txt
done
