#!/bin/bash
# tools/ab_libs.sh: A/B of library builds on one GPU box -- bench.py per workload under each gpurun_ab/lib_<name>.so (TRXSIG_LIB)
# LIBS="base rach" WLS="rach config4" bash tools/ab_libs.sh   (e.g. builds with and without LSO_OFF, see csrc/Makefile)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for L in ${LIBS:-base rach all}; do
  for W in ${WLS:-normal rach config4 config5}; do
    TRXSIG_LIB=$R/gpurun_ab/lib_$L.so timeout -k 10 200 python3 $R/bench.py --workload $W --no-cpu-baseline --no-fresh > /tmp/ab.json 2>/tmp/ab.err || { echo "$L $W failed"; tail -3 /tmp/ab.err; continue; }
    python3 -c "
import json;d=json.load(open('/tmp/ab.json'));print('$L $W', d['value'], d['roofline'].get('kernels_ms'))"
  done
done
