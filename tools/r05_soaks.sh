#!/bin/bash
# tools/r05_soaks.sh: the round's parity campaign (both soft modes) and soaks on the final kernels -> profiles/r05_parity_campaign.txt,
# r05_group_soak.txt, r05_fused_soak.txt
mkdir -p gpurun_out/r05
timeout -k 10 900 python tools/parity_campaign.py 4096 > gpurun_out/r05/parity_campaign.txt 2>&1 || { tail -5 gpurun_out/r05/parity_campaign.txt; exit 1; }
tail -3 gpurun_out/r05/parity_campaign.txt | cut -c1-300
timeout -k 10 600 python tools/group_soak.py 8 > gpurun_out/r05/group_soak.txt 2>&1 || { tail -5 gpurun_out/r05/group_soak.txt; exit 1; }
tail -2 gpurun_out/r05/group_soak.txt | cut -c1-300
timeout -k 10 600 python tools/fused_soak.py 300 > gpurun_out/r05/fused_soak.txt 2>&1 || { tail -5 gpurun_out/r05/fused_soak.txt; exit 1; }
tail -2 gpurun_out/r05/fused_soak.txt | cut -c1-300
