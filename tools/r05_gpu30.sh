#!/bin/bash
mkdir -p gpurun_out/r05
for p in 40 40 250 250 1000; do
python bench.py --steps 20 --warmup 5 --preheat-ms $p --no-cpu-baseline --no-fresh --no-lever 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('preheat $p', d['value'], d['repetitions']['ms_per_step'])"
done
