bash tools/r05_replay.sh && bash tools/r05_replay_probe.sh > gpurun_out/r05/replay_probe_all.txt 2>&1; grep -E "block 0 |group_cache" gpurun_out/r05/replay_probe_all.txt | tail -10
