#!/bin/bash
mkdir -p gpurun_out/r05
for a in "128 8" "128 8" "128 1" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench12.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench12.txt | cut -c1-600
