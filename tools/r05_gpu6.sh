#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py tests/test_gpu_txchain.py -x -q > gpurun_out/r05/gputests_d.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_d.log; tail -12 gpurun_out/r05/gputests_d.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
{
timeout -k 10 120 python tools/group_tx_bench.py 128 8 staged || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 128 8 copy || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 128 1 staged || exit 1
timeout -k 10 120 python tools/group_tx_bench.py 512 8 staged || exit 1
} > gpurun_out/r05/group_tx_bench.txt 2> gpurun_out/r05/group_tx_bench.err
rc2=$?; cat gpurun_out/r05/group_tx_bench.txt | cut -c1-600; tail -3 gpurun_out/r05/group_tx_bench.err
exit $rc2
