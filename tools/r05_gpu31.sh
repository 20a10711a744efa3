#!/bin/bash
bash tools/r05_profiles.sh stats > gpurun_out/r05/profiles_stats2.log 2>&1; grep "^==\|failed" gpurun_out/r05/profiles_stats2.log | head -40
