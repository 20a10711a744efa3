#!/bin/bash
# tools/pmc.sh <tag> [bench args]: rocprofv3 counter passes for our kernels (run on the GPU box via gpurun).
# PMC_PROG=tools/fec_bench.py (or another script under the repo) profiles that program instead of bench.py.
# Counters are collected in separate passes with --kernel-trace only (no sys/hip/hsa tracing).
set -o pipefail
TAG=${1:-pmc}; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --kernel-include-regex "k_(tsc|demod|rach|normal|modulate|resample|rx_resample|channelise|energy|eq_|design_dfe|fec_|unpack|pack|convolve|delay|peak_detect|interpolate|elementwise|decimate|burst_index|group|vector|frequency|add_vector|tx_ring)" --output-format csv -d $OUT/p$i -- \
      python3 $R/${PMC_PROG:-bench.py} $([ -z "$PMC_PROG" ] && echo "--steps 3 --warmup 1 --no-cpu-baseline") "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# the raw per-dispatch tables are large (gpurun brings back at most 64 MiB): keep the summary only unless asked
[ -n "$PMC_KEEP_RAW" ] || rm -rf $OUT/p[0-9]
