#!/bin/bash
# tools/r05_replay_probe.sh: what the state machine's kernels spend their time on (make -C openbts-ttsou_amd/csrc probe_replay: wall_clock64() stamps, printf)
mkdir -p gpurun_out/r05
for w in "--workload config4" "--workload config4 --reference-chain"; do
  TRXSIG_LIB=$PWD/openbts-ttsou_amd/csrc/build_probe/libtrxsig_replayprobe.so timeout -k 10 280 python bench.py $w --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline --no-fresh --no-lever > gpurun_out/r05/replay_probe.out 2> gpurun_out/r05/replay_probe.err
  echo "== $w"; grep -E "replay_wave|group_cache" gpurun_out/r05/replay_probe.out | tail -40
done
