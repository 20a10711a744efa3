#!/bin/bash
# tools/r05_replay.sh: the state machine's wave-per-segment kernel on a GPU box -- the group's tests, then config 4 and the reference chain with
# the form that steps through every slot (TRXSIG_TUNE_GROUP_REPLAY 1, round 4) and with the default, same box, back to back
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_gpu_trxgroup.py tests/test_gpu_config4.py -x -q -m gpu > gpurun_out/r05/replay_tests.txt 2>&1; rc=$?; tail -4 gpurun_out/r05/replay_tests.txt; [ $rc = 0 ] || exit 1
for form in 1 0 1 0; do
  for w in "--workload config4" "--workload config4 --reference-chain"; do
    timeout -k 10 280 python bench.py $w --group-replay $form --no-cpu-baseline > gpurun_out/r05/replay_b.json 2> gpurun_out/r05/replay_b.err || { echo "bench failed"; tail -3 gpurun_out/r05/replay_b.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('gpurun_out/r05/replay_b.json').read().strip().split('\n')[-1]); print('form $form', '$w', d['value'], d['ms_per_step'], d['roofline'].get('kernels_ms'))" | tee -a gpurun_out/r05/replay_ab.txt
  done
done
