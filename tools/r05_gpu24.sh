#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py tests/test_gpu_txchain.py tests/test_gpu_txpath.py -x -q -m gpu > gpurun_out/r05/gputests_w.log 2>&1; rc=$?; tail -3 gpurun_out/r05/gputests_w.log; [ $rc = 0 ] || exit 1
for a in "128 8" "128 8" "128 1" "128 4" "512 8" "512 8" "1024 8"; do timeout -k 10 120 openbts-ttsou_amd/tx_bench $a || exit 1; done > gpurun_out/r05/group_tx_bench_cpp2.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench_cpp2.txt | cut -c50-500
for a in "128 8" "128 8" "512 8"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench24.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench24.txt | cut -c100-420
