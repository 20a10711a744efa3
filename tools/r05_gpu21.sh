#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_trxgroup.py tests/test_gpu_txchain.py tests/test_gpu_txpath.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/r05/gputests_m.log 2>&1; rc=$?; tail -5 gpurun_out/r05/gputests_m.log; [ $rc = 0 ] || exit 1
for a in "128 8" "128 1" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench11.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench11.txt | cut -c1-600
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r05/bench_c4_pin.json 2> gpurun_out/r05/bench_c4_pin.err || exit 1
cut -c1-400 gpurun_out/r05/bench_c4_pin.json
