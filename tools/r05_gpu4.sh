#!/bin/bash
# round 5, fourth GPU session: tolerance tests (incl. the group), the group / config-4 / txgroup suites, the config-4 A/B
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_soft_tolerance.py tests/test_gpu_trxgroup.py tests/test_gpu_trxgroup_tx.py tests/test_gpu_config4.py tests/test_gpu_fullsize.py tests/test_gpu_equalize.py -x -q > gpurun_out/r05/gputests_b.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_b.log; tail -8 gpurun_out/r05/gputests_b.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 300 python tools/config4_ab.py > gpurun_out/r05/config4_ab.txt 2> gpurun_out/r05/config4_ab.err
rc2=$?; cut -c1-420 gpurun_out/r05/config4_ab.txt; tail -3 gpurun_out/r05/config4_ab.err
exit $rc2
