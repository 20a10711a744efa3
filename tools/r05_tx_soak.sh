#!/bin/bash
# tools/r05_tx_soak.sh: the transmit half's cell-by-cell test (tests/test_gpu_trxgroup_tx.py::test_group_transmit_half: the group against
# 128 single objects = std::priority_queue and against the model, deep queues, ties, far bursts) on more random traffic: seeds 1 .. 8
mkdir -p gpurun_out/r05
for seed in 1 2 3 4 5 6 7 8; do
  TRXSIG_TX_SOAK_SEED=$seed timeout -k 10 300 python -m pytest tests/test_gpu_trxgroup_tx.py -q -m gpu -k "test_group_transmit_half" 2>&1 | tail -1 | sed "s/^/seed $seed: /"
done
