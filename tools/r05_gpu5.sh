#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_soft_tolerance.py tests/test_gpu_trxgroup.py -x -q > gpurun_out/r05/gputests_c.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_c.log; tail -8 gpurun_out/r05/gputests_c.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python tools/config4_ab.py > gpurun_out/r05/config4_ab2.txt 2> gpurun_out/r05/config4_ab2.err
rc2=$?; cut -c1-420 gpurun_out/r05/config4_ab2.txt; tail -3 gpurun_out/r05/config4_ab2.err
exit $rc2
