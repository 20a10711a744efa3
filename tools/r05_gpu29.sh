#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05/full_gpu3.txt 2>&1; rc=$?; tail -4 gpurun_out/r05/full_gpu3.txt; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/r05_benches.sh > gpurun_out/r05/benches2.log 2>&1; cat gpurun_out/r05/benches2.log | cut -c1-260
