#!/bin/bash
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/r05/gputests_x.log 2>&1; rc=$?; tail -3 gpurun_out/r05/gputests_x.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline > gpurun_out/r05/bench_c4_split.json 2> gpurun_out/r05/bench_c4_split.err || exit 1
python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_c4_split.json').read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], {k[:20]:round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()})"
