#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py -x -q -m gpu > gpurun_out/r05/gputests_t.log 2>&1; rc=$?; tail -5 gpurun_out/r05/gputests_t.log; [ $rc = 0 ] || exit 1
for a in "128 8" "128 1" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench19.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench19.txt | cut -c1-700
export TMPDIR=/tmp; R=$(pwd); cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05/prof_tx11 -- python3 $R/tools/group_tx_bench.py 128 8 > /dev/null 2> $R/gpurun_out/r05/prof_tx11.log || exit 1
python3 $R/tools/prof_summary.py $R/gpurun_out/r05/prof_tx11 | cut -c1-200
rm -rf $R/gpurun_out/r05/prof_tx11
cd $R
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_txprobe.so timeout -k 10 200 python tools/group_tx_probe.py 128 8 2>&1 | grep -v amdgpu
