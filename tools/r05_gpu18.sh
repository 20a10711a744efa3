#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05/hiptrace
mkdir -p $O
cd /tmp
for M in staged copy; do
timeout -k 10 240 rocprofv3 --hip-runtime-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/$M -- python3 $R/tools/group_tx_bench.py 128 8 $M > $O/$M.json 2> $O/$M.err || exit 1
for f in $(find $O/$M -name "*_stats.csv"); do echo "== $M $f"; head -25 $f | cut -c1-220; done
find $O/$M -name "*trace.csv" -size +20M -delete
done
