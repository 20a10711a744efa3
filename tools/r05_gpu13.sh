#!/bin/bash
mkdir -p gpurun_out/r05
bash tools/r05_profiles.sh pmc > gpurun_out/r05/profiles_pmc.log 2>&1; tail -30 gpurun_out/r05/profiles_pmc.log | cut -c1-160
