#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py -x -q > gpurun_out/r05/gputests_e.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_e.log; tail -12 gpurun_out/r05/gputests_e.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
{
timeout -k 10 120 python tools/group_tx_bench.py 128 8 staged || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 128 8 copy || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 128 1 staged || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 512 8 staged || exit 1
} > gpurun_out/r05/group_tx_bench2.txt 2> gpurun_out/r05/group_tx_bench2.err
rc2=$?; cat gpurun_out/r05/group_tx_bench2.txt | cut -c1-420; tail -3 gpurun_out/r05/group_tx_bench2.err
if [ $rc2 -ne 0 ]; then exit $rc2; fi
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx -o tx -- python3 $GRAFT_REPO_ROOT/tools/group_tx_bench.py 128 8 staged > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r05/prof_tx -name "*kernel_stats*" | head -2; f=$(find gpurun_out/r05/prof_tx -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -14
