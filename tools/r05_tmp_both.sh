bash tools/r05_final.sh 2>&1 | tail -8
bash tools/r05_benches.sh 2>&1 | tail -12
bash tools/r05_replay_prof.sh > gpurun_out/r05/replay_prof.txt 2>&1; grep -c . gpurun_out/r05/replay_prof.txt
bash tools/r05_replay_probe.sh > gpurun_out/r05/replay_probe_all.txt 2>&1; grep -c . gpurun_out/r05/replay_probe_all.txt
