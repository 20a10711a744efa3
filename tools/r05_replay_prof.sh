#!/bin/bash
# tools/r05_replay_prof.sh: kernel times of config 4 and the reference chain with the wave-per-segment state machine (rocprofv3 kernel trace)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05prof
mkdir -p $O
cd /tmp
for W in config4 config4_reference_chain; do
  A="--workload config4"; [ $W = config4_reference_chain ] && A="--workload config4 --reference-chain"
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$W -- python3 $R/bench.py $A --steps 200 --no-cpu-baseline --no-fresh --no-lever > $O/bench_under_rocprof_$W.json 2> $O/stats_$W.err || echo "stats $W failed"
  python3 $R/tools/prof_summary.py $O/stats_$W > $O/kernel_stats_$W.csv
  rm -rf $O/stats_$W
  cat $O/kernel_stats_$W.csv
done
