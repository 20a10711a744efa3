#!/bin/bash
# tools/r05_final.sh: the round's last check on a GPU box -- the whole GPU suite, smoke(), the default bench line, the driver-style one
mkdir -p gpurun_out/r05
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05/full_gpu_final.txt 2>&1; rc=$?; tail -4 gpurun_out/r05/full_gpu_final.txt; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/r05/bench_default_final.json 2> gpurun_out/r05/bench_default_final.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_default_final.json').read().strip().split('\n')[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('read_frac'), d['cpu_baseline']['value'])"
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_driver_final.json 2> gpurun_out/r05/bench_driver_final.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/bench_driver_final.json').read().strip().split('\n')[-1]); print('driver style', d['value'], d['repetitions']['ms_per_step'])"
