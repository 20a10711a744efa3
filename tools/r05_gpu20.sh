#!/bin/bash
mkdir -p gpurun_out/r05
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 8 > gpurun_out/r05/tx_probe.txt 2>&1 || { tail -20 gpurun_out/r05/tx_probe.txt; exit 1; }
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 1 >> gpurun_out/r05/tx_probe.txt 2>&1
grep -v amdgpu gpurun_out/r05/tx_probe.txt
