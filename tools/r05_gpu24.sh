#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py -x -q -m gpu > gpurun_out/r05/gputests_w.log 2>&1; rc=$?; tail -3 gpurun_out/r05/gputests_w.log; [ $rc = 0 ] || exit 1
for a in "128 8" "128 8" "128 1" "128 4" "512 8" "512 8" "1024 8"; do timeout -k 10 120 openbts-ttsou_amd/tx_bench $a || exit 1; done > gpurun_out/r05/group_tx_bench_cpp3.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench_cpp3.txt | cut -c50-420
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 8 2>&1 | grep -v amdgpu | cut -c1-300
