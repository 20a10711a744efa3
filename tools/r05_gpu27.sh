#!/bin/bash
mkdir -p gpurun_out/r05
for a in "128 8" "128 8" "128 1" "128 4" "512 8" "512 8" "1024 8"; do timeout -k 10 120 openbts-ttsou_amd/tx_bench $a || exit 1; done > gpurun_out/r05/group_tx_bench_cpp.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench_cpp.txt | cut -c50-500
