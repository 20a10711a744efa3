#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py -x -q -m gpu > gpurun_out/r05/gputests_z.log 2>&1; rc=$?; tail -3 gpurun_out/r05/gputests_z.log; [ $rc = 0 ] || { grep -n "^E" gpurun_out/r05/gputests_z.log | head -5; exit 1; }
for a in "128 8" "128 8" "128 8" "512 8" "512 8" "128 4"; do timeout -k 10 120 openbts-ttsou_amd/tx_bench $a || exit 1; done 2>&1 | grep -v amdgpu | cut -c50-330
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 8 2>&1 | grep -v amdgpu | cut -c1-800
