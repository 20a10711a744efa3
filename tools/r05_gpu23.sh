#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05/full_gpu2.txt 2>&1; rc=$?; tail -5 gpurun_out/r05/full_gpu2.txt; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for a in "128 8" "128 1" "128 1" "128 2" "128 4" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench20.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench20.txt | cut -c1-330
