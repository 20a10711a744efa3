#!/bin/bash
mkdir -p gpurun_out/r05
bash tools/r05_tx_soak.sh > gpurun_out/r05/tx_soak.txt 2>&1; cat gpurun_out/r05/tx_soak.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/normal_driverstyle.json 2> gpurun_out/r05/normal_driverstyle.err; python3 -c "
import json; d=json.loads(open('gpurun_out/r05/normal_driverstyle.json').read().strip().split('\n')[-1]); print('driver style', d['value'], d['repetitions']['ms_per_step'])"
bash tools/r05_profiles.sh stats > gpurun_out/r05/profiles_stats2.log 2>&1; grep "^==\|failed" gpurun_out/r05/profiles_stats2.log | head -40
