#!/bin/bash
mkdir -p gpurun_out/r05
P=openbts-ttsou_amd/csrc/build_probe
timeout -k 10 600 python -m pytest tests/test_gpu_normal.py tests/test_gpu_normal_fused.py tests/test_gpu_soft_tolerance.py tests/test_gpu_chain.py tests/test_gpu_trxgroup_tx.py tests/test_gpu_txchain.py -x -q > gpurun_out/r05/gputests_f.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_f.log; tail -6 gpurun_out/r05/gputests_f.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
{
timeout -k 10 200 python tools/tol_sweep.py --steps 200 --skip-tuning || exit 1
TRXSIG_LIB=$P/libtrxsig_pk1024.so timeout -k 10 200 python tools/tol_sweep.py --steps 200 --skip-tuning || exit 1
TRXSIG_LIB=$P/libtrxsig_pk256.so timeout -k 10 200 python tools/tol_sweep.py --steps 200 --skip-tuning || exit 1
} > gpurun_out/r05/tol_sweep2.txt 2> gpurun_out/r05/tol_sweep2.err
rc2=$?; cut -c1-330 gpurun_out/r05/tol_sweep2.txt; tail -3 gpurun_out/r05/tol_sweep2.err
if [ $rc2 -ne 0 ]; then exit $rc2; fi
timeout -k 10 120 python tools/group_tx_bench.py 128 8 staged > gpurun_out/r05/group_tx_bench3.txt 2>&1; cut -c1-300 gpurun_out/r05/group_tx_bench3.txt
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx -- python3 $GRAFT_REPO_ROOT/tools/group_tx_bench.py 128 8 staged > $GRAFT_REPO_ROOT/gpurun_out/r05/prof_tx.log 2>&1
cd $GRAFT_REPO_ROOT; python3 tools/prof_summary.py gpurun_out/r05/prof_tx > gpurun_out/r05/kernel_stats_tx.csv; rm -rf gpurun_out/r05/prof_tx; cat gpurun_out/r05/kernel_stats_tx.csv
