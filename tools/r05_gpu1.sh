#!/bin/bash
# round 5, first GPU session: tolerance-mode parity tests, then the arrangement sweep (no GPU step after one that timed out)
mkdir -p gpurun_out/r05
timeout -k 10 420 python -m pytest tests/test_gpu_soft_tolerance.py -x -q > gpurun_out/r05/tol_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r05/tol_tests.log
tail -5 gpurun_out/r05/tol_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 420 python tools/tol_sweep.py --steps 200 > gpurun_out/r05/tol_sweep.txt 2> gpurun_out/r05/tol_sweep.err
rc2=$?
cut -c1-420 gpurun_out/r05/tol_sweep.txt
tail -3 gpurun_out/r05/tol_sweep.err
exit $rc2
