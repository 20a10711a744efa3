#!/bin/bash
mkdir -p gpurun_out/r05
P=openbts-ttsou_amd/csrc/build_probe
timeout -k 10 600 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py -x -q > gpurun_out/r05/gputests_g.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_g.log; tail -6 gpurun_out/r05/gputests_g.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
{
for r in 1 2; do
timeout -k 10 200 python tools/tol_sweep.py --steps 300 --skip-tuning || exit 1
TRXSIG_LIB=$P/libtrxsig_cw0.so timeout -k 10 200 python tools/tol_sweep.py --steps 300 --skip-tuning || exit 1
done
} > gpurun_out/r05/tol_sweep3.txt 2> gpurun_out/r05/tol_sweep3.err
rc2=$?; grep -v beside gpurun_out/r05/tol_sweep3.txt | cut -c1-200; tail -3 gpurun_out/r05/tol_sweep3.err
if [ $rc2 -ne 0 ]; then exit $rc2; fi
for a in "128 8" "128 1" "512 8"; do timeout -k 10 120 python tools/group_tx_bench.py $a staged || exit 1; done > gpurun_out/r05/group_tx_bench4.txt 2>&1; cut -c1-330 gpurun_out/r05/group_tx_bench4.txt
