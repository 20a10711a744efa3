#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py tests/test_gpu_trxgroup.py tests/test_gpu_equalize.py -x -q > gpurun_out/r05/gputests_j.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_j.log; tail -4 gpurun_out/r05/gputests_j.log
if [ $rc -ne 0 ]; then exit $rc; fi
TRXSIG_LIB=openbts-ttsou_amd/csrc/build_probe/libtrxsig_d4probe.so timeout -k 10 200 python tools/dfe4_probe.py > gpurun_out/r05/dfe4_probe.txt 2>&1; cat gpurun_out/r05/dfe4_probe.txt | grep -v amdgpu
for a in "128 8" "512 8"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench6.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench6.txt | cut -c1-260
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx2 -- python3 $GRAFT_REPO_ROOT/tools/group_tx_bench.py 128 8 staged > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx2.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r05/prof_tx2 -name "*stats*.csv" | head; for f in $(find gpurun_out/r05/prof_tx2 -name "*memory_copy_stats.csv" -o -name "*kernel_stats.csv"); do echo "== $f"; cut -c1-170 $f | head -8; done; rm -rf gpurun_out/r05/prof_tx2
