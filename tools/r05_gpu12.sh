#!/bin/bash
mkdir -p gpurun_out/r05
bash tools/r05_profiles.sh stats > gpurun_out/r05/profiles_stats.log 2>&1; tail -40 gpurun_out/r05/profiles_stats.log | cut -c1-160
