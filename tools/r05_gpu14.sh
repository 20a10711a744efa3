#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup_tx.py tests/test_gpu_udp.py tests/test_gpu_soft_tolerance.py tests/test_gpu_rach.py tests/test_gpu_config4.py -x -q > gpurun_out/r05/gputests_i.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r05/gputests_i.log; tail -6 gpurun_out/r05/gputests_i.log
if [ $rc -ne 0 ]; then exit $rc; fi
for a in "128 8" "128 1" "512 8" "128 8 copy"; do timeout -k 10 120 python tools/group_tx_bench.py $a || exit 1; done > gpurun_out/r05/group_tx_bench5.txt 2>&1; grep -v amdgpu gpurun_out/r05/group_tx_bench5.txt | cut -c1-330
timeout -k 10 280 python bench.py --workload rach --no-cpu-baseline > gpurun_out/r05/bench_rach_b.json 2> gpurun_out/r05/bench_rach_b.err; python3 -c "
import json; d=json.load(open('gpurun_out/r05/bench_rach_b.json')); print('rach', d['value'], d['roofline']['kernels_ms'], (d.get('other_soft_mode') or {}).get('value'), (d.get('other_soft_mode') or {}).get('kernels_ms'))"
