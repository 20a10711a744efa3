#!/bin/bash
# round 5: the whole GPU suite, the bench lines, then the rocprofv3 evidence (stats + counter passes)
mkdir -p gpurun_out/r05
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r05/full_gpu.txt 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/full_gpu.txt; tail -5 gpurun_out/r05/full_gpu.txt
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/r05_benches.sh > gpurun_out/r05/benches.log 2>&1; tail -12 gpurun_out/r05/benches.log
