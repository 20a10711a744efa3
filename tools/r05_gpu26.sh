#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py -x -q -m gpu > gpurun_out/r05/gputests_v.log 2>&1; rc=$?; tail -5 gpurun_out/r05/gputests_v.log; [ $rc = 0 ] || exit 1
